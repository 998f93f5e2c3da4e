// one tile configuration of the conv engine (see conv_engine_impl.h)
#include "conv_engine_impl.h"

int ag_conv_cfg_2114(ConvP& p, hipStream_t st) { return launch_cfg<2, 1, 1, 4>(p, st); }
