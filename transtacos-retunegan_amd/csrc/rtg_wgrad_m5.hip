// rtg_wgrad_m5.hip — wgrad kernel instances of addressing mode 1 (continuous virtual sequence, 1-D rows) with bf16 operands (mode bit 2)
#include "rtg_wgrad_kernel.h"

RTG_WGRAD_DEFINE_MODE(5)
