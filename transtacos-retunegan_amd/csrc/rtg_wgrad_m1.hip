// rtg_wgrad_m1.hip — wgrad kernel instances of addressing mode 1 (continuous tiling, 1-D rows)
#include "rtg_wgrad_kernel.h"

RTG_WGRAD_DEFINE_MODE(1)
RTG_WGRAD_DEFINE_GROUP(1)
