// rtg_conv1d_t3221.hip — conv1d_mfma_kernel instances of block shape TM=32, MT=2, NT=1
#include "rtg_conv1d_kernel.h"

RTG_CONV_DEFINE(32, 2, 1)
