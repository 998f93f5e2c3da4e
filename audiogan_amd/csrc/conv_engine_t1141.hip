// one tile configuration of the conv engine (see conv_engine_impl.h)
#include "conv_engine_impl.h"

int ag_conv_cfg_1141(ConvP& p, hipStream_t st) { return launch_cfg<1, 1, 4, 1>(p, st); }
