// one tile configuration of the conv engine (see conv_engine_impl.h)
#include "conv_engine_impl.h"

int ag_conv_cfg_1122(ConvP& p, hipStream_t st) { return launch_cfg<1, 1, 2, 2>(p, st); }
