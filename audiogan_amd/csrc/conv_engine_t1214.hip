// one tile configuration of the conv engine (see conv_engine_impl.h)
#include "conv_engine_impl.h"

int ag_conv_cfg_1214(ConvP& p, hipStream_t st) { return launch_cfg<1, 2, 1, 4>(p, st); }
