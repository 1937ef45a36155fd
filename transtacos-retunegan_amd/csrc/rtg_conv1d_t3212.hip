// rtg_conv1d_t3212.hip — conv1d_mfma_kernel instances of block shape TM=32, MT=1, NT=2
#include "rtg_conv1d_kernel.h"

RTG_CONV_DEFINE(32, 1, 2)
