// rtg_dconv_bf.hip — the dense-layer conv kernel (rtg_dconv.hip, rtg_dconv_kernel.h) with bf16 operand fragments on fp32
// tensors (hparam.compute_dtype = 'bf16', BASELINE configs[2]).  A translation unit of its own: its instances compile side by
// side with the fp32 ones of rtg_dconv.hip and the bf16-tensor ones of rtg_dconv_io{1,2,3}.hip.
#include "rtg_dconv_kernel.h"

int rtg_dconv_launch_bf(const rtg_dc::DArgs& a, int si, int nt16, int S, int K, bool two_d, unsigned blocks, size_t lds_bytes,
                        hipStream_t s) {
  return rtg_dc::launch_shape<true, 0>(a, si, nt16, S, K, two_d, blocks, lds_bytes, s);
}
