// rtg_wgrad_m6.hip — wgrad kernel instances of addressing mode 2 (per-clip tiling, 2-D rows) with bf16 operands (mode bit 2)
#include "rtg_wgrad_kernel.h"

RTG_WGRAD_DEFINE_MODE(6)
