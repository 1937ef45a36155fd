// rtg_conv1d_t1614.hip — conv1d_mfma_kernel instances of block shape TM=16, MT=1, NT=4
#include "rtg_conv1d_kernel.h"

RTG_CONV_DEFINE(16, 1, 4)
