// rtg_wgrad_m0.hip — wgrad kernel instances of addressing mode 0 (per-clip tiling, 1-D rows)
#include "rtg_wgrad_kernel.h"

RTG_WGRAD_DEFINE_MODE(0)
RTG_WGRAD_DEFINE_GROUP(0)
