// rtg_wgrad_m2.hip — wgrad kernel instances of addressing mode 2 (per-clip tiling, 2-D rows)
#include "rtg_wgrad_kernel.h"

RTG_WGRAD_DEFINE_MODE(2)
