// rtg_wgrad_m7.hip — wgrad kernel instances of addressing mode 3 (continuous virtual sequence, 2-D rows) with bf16 operands (mode bit 2)
#include "rtg_wgrad_kernel.h"

RTG_WGRAD_DEFINE_MODE(7)
