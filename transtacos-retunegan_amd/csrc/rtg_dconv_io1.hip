// rtg_dconv_io1.hip — the dense-layer conv kernel (rtg_dconv.hip, rtg_dconv_kernel.h) on bf16 feature maps in HBM:
// the instances with RtgConv1dDesc.io_bf16 == 1 (x bf16, out fp32).  A translation unit of its own: the
// block shapes x (stride, taps, 1-D / 2-D) instances of one tensor-type combination compile side by side with the others.
#include "rtg_dconv_kernel.h"

int rtg_dconv_launch_io1(const rtg_dc::DArgs& a, int si, int nt16, int S, int K, bool two_d, unsigned blocks, size_t lds_bytes,
                         hipStream_t s) {
  return rtg_dc::launch_shape<true, 1>(a, si, nt16, S, K, two_d, blocks, lds_bytes, s);
}
