// rtg_wgrad_m3.hip — wgrad kernel instances of addressing mode 3 (continuous tiling, 2-D rows)
#include "rtg_wgrad_kernel.h"

RTG_WGRAD_DEFINE_MODE(3)
