// rtg_elem.hip — the HBM-bound element-wise / reduction kernels of the RetuneGAN train step (see include/rtg.h):
// GaussianNoise, axpby, leaky-relu backward, AvgPool1d(4,2,1), DiscriminatorP fold, the scalar losses (multi-tensor,
// one launch for a whole list of feature maps / logits) and the flat fused AdamW.
#include "rtg_common.h"

namespace {

constexpr int MAX_GRID = 2048;

inline int grid_for(long long n, int per_thread = 4) {
  long long g = (n + (long long)RTG_THREADS * per_thread - 1) / ((long long)RTG_THREADS * per_thread);
  if (g < 1) g = 1;
  if (g > MAX_GRID) g = MAX_GRID;
  return (int)g;
}

// counter-based uniform [0,1): splitmix64 finaliser of (seed, index); 24 mantissa bits
__device__ __forceinline__ float uniform01(unsigned long long seed, unsigned long long idx) {
  unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (idx + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (float)(z >> 40) * (1.0f / 16777216.0f);
}

__device__ __forceinline__ unsigned long long mix_salt(unsigned long long seed, const float* salt) {
  if (!salt) return seed;
  return seed ^ (0xD1B54A32D192ED03ull * (unsigned long long)(__float_as_uint(*salt) + 1u));
}

// The element-wise kernels move 16 bytes per lane and access (Guideline 13 of the CDNA guide: a 4-byte-per-lane stream tops
// out far below the HBM rate): VEC = 4 when every pointer is 16-byte aligned and n % 4 == 0 (all feature maps of the
// path), else the scalar form.  Element i draws the same random number either way (the counter is the element index).
template <int VEC>
__global__ __launch_bounds__(RTG_THREADS) void noise_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                const float* __restrict__ u_in, float* __restrict__ out,
                                                                long long n, float slope, unsigned long long seed,
                                                                const float* salt) {
  const float wv = *w;
  const unsigned long long sd = mix_salt(seed, salt);
  const long long nv = n / VEC;
  for (long long i = (long long)blockIdx.x * RTG_THREADS + threadIdx.x; i < nv; i += (long long)gridDim.x * RTG_THREADS) {
    float xv[VEC], uv[VEC], ov[VEC];
    if constexpr (VEC == 4) {
      const f32x4 t = reinterpret_cast<const f32x4*>(x)[i];
      xv[0] = t.x; xv[1] = t.y; xv[2] = t.z; xv[3] = t.w;
      if (u_in) {
        const f32x4 u4 = reinterpret_cast<const f32x4*>(u_in)[i];
        uv[0] = u4.x; uv[1] = u4.y; uv[2] = u4.z; uv[3] = u4.w;
      }
    } else {
      xv[0] = x[i];
      if (u_in) uv[0] = u_in[i];
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const float u = u_in ? uv[e] : uniform01(sd, (unsigned long long)(i * VEC + e));
      ov[e] = rtg_lrelu(xv[e] + u * wv, slope);
    }
    if constexpr (VEC == 4) reinterpret_cast<f32x4*>(out)[i] = f32x4{ov[0], ov[1], ov[2], ov[3]};
    else out[i] = ov[0];
  }
}

template <int VEC>
__global__ __launch_bounds__(RTG_THREADS) void noise_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                const float* __restrict__ u_in,
                                                                const float* __restrict__ dy, float* __restrict__ dx,
                                                                float* __restrict__ dw_part, long long n, float slope,
                                                                unsigned long long seed, const float* salt) {
  __shared__ float red[4];
  const float wv = *w;
  const unsigned long long sd = mix_salt(seed, salt);
  float acc = 0.f;
  const long long nv = n / VEC;
  for (long long i = (long long)blockIdx.x * RTG_THREADS + threadIdx.x; i < nv; i += (long long)gridDim.x * RTG_THREADS) {
    float xv[VEC], uv[VEC], gv[VEC], ov[VEC];
    if constexpr (VEC == 4) {
      const f32x4 t = reinterpret_cast<const f32x4*>(x)[i], g4 = reinterpret_cast<const f32x4*>(dy)[i];
      xv[0] = t.x; xv[1] = t.y; xv[2] = t.z; xv[3] = t.w;
      gv[0] = g4.x; gv[1] = g4.y; gv[2] = g4.z; gv[3] = g4.w;
      if (u_in) {
        const f32x4 u4 = reinterpret_cast<const f32x4*>(u_in)[i];
        uv[0] = u4.x; uv[1] = u4.y; uv[2] = u4.z; uv[3] = u4.w;
      }
    } else {
      xv[0] = x[i];
      gv[0] = dy[i];
      if (u_in) uv[0] = u_in[i];
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const float u = u_in ? uv[e] : uniform01(sd, (unsigned long long)(i * VEC + e));
      const float pre = xv[e] + u * wv;
      ov[e] = gv[e] * (pre > 0.f ? 1.f : slope);
      acc += ov[e] * u;
    }
    if constexpr (VEC == 4) reinterpret_cast<f32x4*>(dx)[i] = f32x4{ov[0], ov[1], ov[2], ov[3]};
    else dx[i] = ov[0];
  }
  acc = rtg_block_sum(acc, red);
  if (threadIdx.x == 0) dw_part[blockIdx.x] = acc;
}

// *acc += sum of part[0 .. n): element t, t + 256, ... per thread, then the block tree — a fixed order.  One block: the
// second launch of rtg_noise_lrelu_bwd_acc (a reduction INSIDE the first launch — the block that arrives last adds the
// partials — needs a device-scope release per block, i.e. a write-back of the XCD's L2 while the kernel streams its 33 MB
// of output through it: measured 15 -> 93 us per launch)
__global__ __launch_bounds__(RTG_THREADS) void reduce_acc_kernel(const float* __restrict__ part, int n, float* __restrict__ acc) {
  __shared__ float red[4];
  float t = 0.f;
  for (int i = threadIdx.x; i < n; i += RTG_THREADS) t += part[i];
  t = rtg_block_sum(t, red);
  if (threadIdx.x == 0) *acc += t;
}

template <int VEC>
__global__ __launch_bounds__(RTG_THREADS) void axpby_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                            float* __restrict__ out, long long n, float alpha,
                                                            float beta, int accumulate) {
  const long long nv = n / VEC;
  for (long long i = (long long)blockIdx.x * RTG_THREADS + threadIdx.x; i < nv; i += (long long)gridDim.x * RTG_THREADS) {
    if constexpr (VEC == 4) {
      f32x4 v = reinterpret_cast<const f32x4*>(a)[i] * alpha;
      if (b) v += reinterpret_cast<const f32x4*>(b)[i] * beta;        // (separate multiply and add, like the scalar form)
      if (accumulate) v += reinterpret_cast<const f32x4*>(out)[i];
      reinterpret_cast<f32x4*>(out)[i] = v;
    } else {
      float v = alpha * a[i];
      if (b) v += beta * b[i];
      if (accumulate) v += out[i];
      out[i] = v;
    }
  }
}

// sum over (clip, position) per channel of x [B, C, L], in two fixed-order stages: block (c, s) sums the clips
// b = s, s + S, ... of channel c into ws[c * S + s] (one block per channel left the chip to 16-128 workgroups: 113 us for
// 18 MB), then one wave per channel adds the S partials in ascending order to out[c].
constexpr int CSUM_SLICES = 32;
__global__ __launch_bounds__(RTG_THREADS) void channel_sum_stage1(const float* __restrict__ x, float* __restrict__ ws,
                                                                  int B, int C, int L) {
  __shared__ float red[4];
  const int c = blockIdx.x, s = blockIdx.y;
  float acc = 0.f;
  const bool vec = (L & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
  for (int b = s; b < B; b += CSUM_SLICES) {
    const float* xr = x + ((size_t)b * C + c) * L;
    if (vec) {
      for (int i = threadIdx.x; i < (L >> 2); i += RTG_THREADS) {
        const f32x4 v = reinterpret_cast<const f32x4*>(xr)[i];
        acc += (v.x + v.y) + (v.z + v.w);
      }
    } else {
      for (int i = threadIdx.x; i < L; i += RTG_THREADS) acc += xr[i];
    }
  }
  acc = rtg_block_sum(acc, red);
  if (threadIdx.x == 0) ws[c * CSUM_SLICES + s] = acc;
}
__global__ __launch_bounds__(RTG_THREADS) void channel_sum_stage2(const float* __restrict__ ws, float* __restrict__ out, int C) {
  const int c = blockIdx.x * RTG_THREADS + threadIdx.x;
  if (c >= C) return;
  float acc = 0.f;
#pragma unroll
  for (int s = 0; s < CSUM_SLICES; ++s) acc += ws[c * CSUM_SLICES + s];
  out[c] += acc;
}

template <int VEC>
__global__ __launch_bounds__(RTG_THREADS) void lrelu_bwd_kernel(const float* __restrict__ dy,
                                                                const float* __restrict__ ref, float* __restrict__ dx,
                                                                long long n, float slope) {
  const long long nv = n / VEC;
  for (long long i = (long long)blockIdx.x * RTG_THREADS + threadIdx.x; i < nv; i += (long long)gridDim.x * RTG_THREADS) {
    if constexpr (VEC == 4) {
      const f32x4 g = reinterpret_cast<const f32x4*>(dy)[i], r = reinterpret_cast<const f32x4*>(ref)[i];
      reinterpret_cast<f32x4*>(dx)[i] = f32x4{g.x * (r.x > 0.f ? 1.f : slope), g.y * (r.y > 0.f ? 1.f : slope),
                                              g.z * (r.z > 0.f ? 1.f : slope), g.w * (r.w > 0.f ? 1.f : slope)};
    } else {
      dx[i] = dy[i] * (ref[i] > 0.f ? 1.f : slope);
    }
  }
}

// bf16 feature maps in HBM: the conversions at the edge of the kernels that read / write them natively (8 elements per thread
// and pass where the pointers allow 16-byte accesses on the bf16 side)
typedef unsigned rtg_u32x4e __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(RTG_THREADS) void bf16_encode_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst,
                                                                  long long n, float slope, int vec) {
  auto enc = [&](float v) __attribute__((always_inline)) {
    return (unsigned)__builtin_bit_cast(unsigned short, (__bf16)(v > 0.f ? v : v * slope));
  };
  long long done = 0;
  if (vec) {
    const long long n8 = n >> 3;
    for (long long i = (long long)blockIdx.x * RTG_THREADS + threadIdx.x; i < n8; i += (long long)gridDim.x * RTG_THREADS) {
      const f32x4 a = reinterpret_cast<const f32x4*>(src)[2 * i], b = reinterpret_cast<const f32x4*>(src)[2 * i + 1];
      reinterpret_cast<rtg_u32x4e*>(dst)[i] = rtg_u32x4e{enc(a.x) | (enc(a.y) << 16), enc(a.z) | (enc(a.w) << 16),
                                                         enc(b.x) | (enc(b.y) << 16), enc(b.z) | (enc(b.w) << 16)};
    }
    done = n8 << 3;
  }
  for (long long i = done + (long long)blockIdx.x * RTG_THREADS + threadIdx.x; i < n; i += (long long)gridDim.x * RTG_THREADS)
    dst[i] = (unsigned short)enc(src[i]);
}

__global__ __launch_bounds__(RTG_THREADS) void bf16_decode_kernel(const unsigned short* __restrict__ src, float* __restrict__ dst,
                                                                  long long n, float inv_slope, int vec) {
  auto dec = [&](unsigned h) __attribute__((always_inline)) {
    const float v = __builtin_bit_cast(float, h << 16);
    return v > 0.f ? v : v * inv_slope;
  };
  long long done = 0;
  if (vec) {
    const long long n8 = n >> 3;
    for (long long i = (long long)blockIdx.x * RTG_THREADS + threadIdx.x; i < n8; i += (long long)gridDim.x * RTG_THREADS) {
      const rtg_u32x4e v = reinterpret_cast<const rtg_u32x4e*>(src)[i];
      reinterpret_cast<f32x4*>(dst)[2 * i] = f32x4{dec(v.x & 0xffffu), dec(v.x >> 16), dec(v.y & 0xffffu), dec(v.y >> 16)};
      reinterpret_cast<f32x4*>(dst)[2 * i + 1] = f32x4{dec(v.z & 0xffffu), dec(v.z >> 16), dec(v.w & 0xffffu), dec(v.w >> 16)};
    }
    done = n8 << 3;
  }
  for (long long i = done + (long long)blockIdx.x * RTG_THREADS + threadIdx.x; i < n; i += (long long)gridDim.x * RTG_THREADS)
    dst[i] = dec(src[i]);
}

__global__ __launch_bounds__(RTG_THREADS) void avgpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                                  int rows, int L) {
  const int Lo = L / 2;
  const long long n = (long long)rows * Lo;
  for (long long i = (long long)blockIdx.x * RTG_THREADS + threadIdx.x; i < n; i += (long long)gridDim.x * RTG_THREADS) {
    const int r = (int)(i / Lo), o = (int)(i - (long long)r * Lo);
    const float* xr = x + (size_t)r * L;
    const int j = 2 * o - 1;
    float s = xr[j + 1] + xr[j + 2];
    if (j >= 0) s += xr[j];
    if (j + 3 < L) s += xr[j + 3];
    out[i] = 0.25f * s;                     // count_include_pad = True (nn.AvgPool1d default)
  }
}

__global__ __launch_bounds__(RTG_THREADS) void avgpool_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx,
                                                                  int rows, int L) {
  const int Lo = L / 2;
  const long long n = (long long)rows * L;
  for (long long i = (long long)blockIdx.x * RTG_THREADS + threadIdx.x; i < n; i += (long long)gridDim.x * RTG_THREADS) {
    const int r = (int)(i / L), j = (int)(i - (long long)r * L);
    const float* g = dy + (size_t)r * Lo;
    // outputs o with 2o-1 <= j <= 2o+2
    const int o_hi = (j + 1) >> 1, o_lo = o_hi - 1;
    float s = 0.f;
    if (o_lo >= 0 && o_lo < Lo) s += g[o_lo];
    if (o_hi < Lo) s += g[o_hi];
    dx[i] = 0.25f * s;
  }
}

__global__ __launch_bounds__(RTG_THREADS) void fold_fwd_kernel(const float* __restrict__ y, float* __restrict__ out,
                                                               int B, int T, int p, int H) {
  const long long n = (long long)B * p * H;
  for (long long i = (long long)blockIdx.x * RTG_THREADS + threadIdx.x; i < n; i += (long long)gridDim.x * RTG_THREADS) {
    const int h = (int)(i % H);
    const long long bw = i / H;
    const int w = (int)(bw % p), b = (int)(bw / p);
    int t = h * p + w;
    if (t >= T) t = 2 * (T - 1) - t;
    out[i] = y[(size_t)b * T + t];
  }
}

__global__ __launch_bounds__(RTG_THREADS) void fold_bwd_kernel(const float* __restrict__ dout, float* __restrict__ dy,
                                                               int B, int T, int p, int H) {
  const long long n = (long long)B * T;
  const int Tp = H * p;
  for (long long i = (long long)blockIdx.x * RTG_THREADS + threadIdx.x; i < n; i += (long long)gridDim.x * RTG_THREADS) {
    const int b = (int)(i / T), t = (int)(i - (long long)b * T);
    const float* g = dout + (size_t)b * p * H;
    float s = g[(size_t)(t % p) * H + t / p];
    const int tr = 2 * (T - 1) - t;          // padded position that reflects onto t
    if (tr >= T && tr < Tp) s += g[(size_t)(tr % p) * H + tr / p];
    dy[i] = s;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// multi-tensor scalar losses
// ---------------------------------------------------------------------------------------------------------------
typedef unsigned rtg_u32x4 __attribute__((ext_vector_type(4)));
struct LossJobs {
  int n_jobs;
  RtgLossJob job[RTG_MAX_LOSS_JOBS];
};
constexpr int LOSS_GX = 64;      // partial sums per job

__device__ __forceinline__ float loss_term(int kind, float a, float b, float target) {
  if (kind == RTG_LOSS_L1) return fabsf(a - b);
  if (kind == RTG_LOSS_L1_L1LOG) return fabsf(a - b) + fabsf(logf(a) - logf(b));
  const float e = target - (kind == RTG_LOSS_MSE_REL ? a - b : a);      // RTG_LOSS_MSE_TARGET / _REL
  return e * e;
}

__global__ __launch_bounds__(RTG_THREADS) void loss_fwd_kernel(int kind, const LossJobs jobs, float* __restrict__ ws) {
  __shared__ float red[4];
  const RtgLossJob j = jobs.job[blockIdx.y];
  float acc = 0.f;
  // 16-byte loads where the job's tensors allow them (a feature map of 150-300 MB is walked by LOSS_GX blocks: with 4-byte
  // loads a thread had one request in flight per iteration, 1.8 TB/s); the order of the additions is fixed either way
  const bool vec = ((reinterpret_cast<uintptr_t>(j.a) | reinterpret_cast<uintptr_t>(j.b)) & 15) == 0;
  long long done = 0;
  if (kind == RTG_LOSS_L1_ENC) {
    // bf16 feature maps (leaky-relu encoded, slope = j.target): 8 elements per 16-byte load
    const float inv = 1.f / j.target;
    const unsigned short* pa = reinterpret_cast<const unsigned short*>(j.a);
    const unsigned short* pb = reinterpret_cast<const unsigned short*>(j.b);
    auto dec = [&](unsigned h) __attribute__((always_inline)) {
      const float v = __builtin_bit_cast(float, h << 16);
      return v > 0.f ? v : v * inv;
    };
    if (vec) {
      const long long n8 = j.n >> 3;
      for (long long i = (long long)blockIdx.x * RTG_THREADS + threadIdx.x; i < n8; i += (long long)LOSS_GX * RTG_THREADS) {
        const rtg_u32x4 va = reinterpret_cast<const rtg_u32x4*>(pa)[i], vb = reinterpret_cast<const rtg_u32x4*>(pb)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e)
          acc += fabsf(dec(va[e] & 0xffffu) - dec(vb[e] & 0xffffu)) + fabsf(dec(va[e] >> 16) - dec(vb[e] >> 16));
      }
      done = n8 << 3;
    }
    for (long long i = done + (long long)blockIdx.x * RTG_THREADS + threadIdx.x; i < j.n; i += (long long)LOSS_GX * RTG_THREADS)
      acc += fabsf(dec(pa[i]) - dec(pb[i]));
    done = j.n;
  } else if (vec) {
    const long long n4 = j.n >> 2;
    const f32x4* a4 = reinterpret_cast<const f32x4*>(j.a);
    const f32x4* b4 = reinterpret_cast<const f32x4*>(j.b);
    for (long long i = (long long)blockIdx.x * RTG_THREADS + threadIdx.x; i < n4; i += (long long)LOSS_GX * RTG_THREADS) {
      const f32x4 va = a4[i];
      const f32x4 vb = j.b ? b4[i] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int e = 0; e < 4; ++e) acc += loss_term(kind, va[e], vb[e], j.target);
    }
    done = n4 << 2;
  }
  for (long long i = done + (long long)blockIdx.x * RTG_THREADS + threadIdx.x; i < j.n; i += (long long)LOSS_GX * RTG_THREADS)
    acc += loss_term(kind, j.a[i], j.b ? j.b[i] : 0.f, j.target);
  acc = rtg_block_sum(acc, red);
  if (threadIdx.x == 0) ws[blockIdx.y * LOSS_GX + blockIdx.x] = acc * (j.w / (float)j.n);
}

__global__ __launch_bounds__(64) void loss_finish_kernel(int n_part, const float* __restrict__ ws,
                                                         float* __restrict__ loss_out) {
  float acc = 0.f;
  for (int i = threadIdx.x; i < n_part; i += 64) acc += ws[i];
  acc = rtg_wave_sum(acc);
  if (threadIdx.x == 0) *loss_out += acc;
}

__global__ __launch_bounds__(RTG_THREADS) void loss_bwd_kernel(int kind, const LossJobs jobs,
                                                               const float* __restrict__ gscale) {
  const RtgLossJob j = jobs.job[blockIdx.y];
  const float k = (gscale ? *gscale : 1.f) * j.w / (float)j.n;
  auto grads = [&](float a, float b, float& ga, float& gb) __attribute__((always_inline)) {
    if (kind == RTG_LOSS_L1) {
      const float s = (a > b) ? 1.f : ((a < b) ? -1.f : 0.f);
      ga = s; gb = -s;
    } else if (kind == RTG_LOSS_L1_L1LOG) {
      const float s = (a > b) ? 1.f : ((a < b) ? -1.f : 0.f);
      const float la = logf(a), lb = logf(b);
      const float sl = (la > lb) ? 1.f : ((la < lb) ? -1.f : 0.f);
      ga = s + sl / a; gb = -s - sl / b;
    } else if (kind == RTG_LOSS_MSE_REL) {
      ga = -2.f * (j.target - (a - b)); gb = 0.f;      // b is detached in the reference (loss.py:116,136)
    } else {
      ga = -2.f * (j.target - a); gb = 0.f;
    }
  };
  long long done = 0;
  if (kind == RTG_LOSS_L1_ENC) {
    // d |dec(a) - dec(b)| / d dec(a) = sign; gradients stored as bf16 (+-k or 0: exact up to the rounding of k)
    const float inv = 1.f / j.target;
    const unsigned short* pa = reinterpret_cast<const unsigned short*>(j.a);
    const unsigned short* pb = reinterpret_cast<const unsigned short*>(j.b);
    unsigned short* da = reinterpret_cast<unsigned short*>(j.da);
    unsigned short* db = reinterpret_cast<unsigned short*>(j.db);
    const unsigned kp = (unsigned)__builtin_bit_cast(unsigned short, (__bf16)k), kn = kp ^ 0x8000u;
    auto dec = [&](unsigned h) __attribute__((always_inline)) {
      const float v = __builtin_bit_cast(float, h << 16);
      return v > 0.f ? v : v * inv;
    };
    auto sg = [&](unsigned ha, unsigned hb) __attribute__((always_inline)) {       // bf16 bits of k * sign(dec(a) - dec(b))
      const float a = dec(ha), b = dec(hb);
      return a > b ? kp : (a < b ? kn : 0u);
    };
    if (((reinterpret_cast<uintptr_t>(j.a) | reinterpret_cast<uintptr_t>(j.b) | reinterpret_cast<uintptr_t>(j.da) |
          reinterpret_cast<uintptr_t>(j.db)) & 15) == 0) {
      const long long n8 = j.n >> 3;
      for (long long i = (long long)blockIdx.x * RTG_THREADS + threadIdx.x; i < n8; i += (long long)gridDim.x * RTG_THREADS) {
        const rtg_u32x4 va = reinterpret_cast<const rtg_u32x4*>(pa)[i], vb = reinterpret_cast<const rtg_u32x4*>(pb)[i];
        rtg_u32x4 oa, ob;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const unsigned lo = sg(va[e] & 0xffffu, vb[e] & 0xffffu), hi = sg(va[e] >> 16, vb[e] >> 16);
          oa[e] = lo | (hi << 16);
          ob[e] = (lo ? lo ^ 0x8000u : 0u) | ((hi ? hi ^ 0x8000u : 0u) << 16);
        }
        if (da) reinterpret_cast<rtg_u32x4*>(da)[i] = oa;
        if (db) reinterpret_cast<rtg_u32x4*>(db)[i] = ob;
      }
      done = n8 << 3;
    }
    for (long long i = done + (long long)blockIdx.x * RTG_THREADS + threadIdx.x; i < j.n; i += (long long)gridDim.x * RTG_THREADS) {
      const unsigned g = sg(pa[i], pb[i]);
      if (da) da[i] = (unsigned short)g;
      if (db) db[i] = (unsigned short)(g ? g ^ 0x8000u : 0u);
    }
    return;
  }
  if (((reinterpret_cast<uintptr_t>(j.a) | reinterpret_cast<uintptr_t>(j.b) | reinterpret_cast<uintptr_t>(j.da) |
        reinterpret_cast<uintptr_t>(j.db)) & 15) == 0) {
    const long long n4 = j.n >> 2;
    for (long long i = (long long)blockIdx.x * RTG_THREADS + threadIdx.x; i < n4; i += (long long)gridDim.x * RTG_THREADS) {
      const f32x4 va = reinterpret_cast<const f32x4*>(j.a)[i];
      const f32x4 vb = j.b ? reinterpret_cast<const f32x4*>(j.b)[i] : f32x4{0.f, 0.f, 0.f, 0.f};
      float oa[4], ob[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float ga, gb;
        grads(va[e], vb[e], ga, gb);
        oa[e] = k * ga; ob[e] = k * gb;
      }
      if (j.da) reinterpret_cast<f32x4*>(j.da)[i] = f32x4{oa[0], oa[1], oa[2], oa[3]};
      if (j.db) reinterpret_cast<f32x4*>(j.db)[i] = f32x4{ob[0], ob[1], ob[2], ob[3]};
    }
    done = n4 << 2;
  }
  for (long long i = done + (long long)blockIdx.x * RTG_THREADS + threadIdx.x; i < j.n; i += (long long)gridDim.x * RTG_THREADS) {
    const float a = j.a[i], b = j.b ? j.b[i] : 0.f;
    float ga, gb;
    grads(a, b, ga, gb);
    if (j.da) j.da[i] = k * ga;
    if (j.db) j.db[i] = k * gb;
  }
}

// dynamic loss: one wavefront per (row, window)
__device__ __forceinline__ void wave_argmax(float& v, int& idx) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float ov = __shfl_xor(v, off, 64);
    const int oi = __shfl_xor(idx, off, 64);
    if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
  }
}

template <bool BWD, bool ENV>      // ENV: envelope_loss (two separate |max - max| terms) instead of dynamic_loss
__global__ __launch_bounds__(RTG_THREADS) void dyn_kernel(const float* __restrict__ y, const float* __restrict__ g,
                                                          int rows, int L, int k, float scale,
                                                          const float* __restrict__ gscale, float* __restrict__ ws,
                                                          float* __restrict__ dg) {
  const int W = L / k;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nwin = rows * W;
  float total = 0.f;
  for (int wi = blockIdx.x * 4 + wave; wi < nwin; wi += gridDim.x * 4) {
    const int r = wi / W, w0 = (wi - r * W) * k;
    const float* yr = y + (size_t)r * L + w0;
    const float* gr = g + (size_t)r * L + w0;
    float ymx = -INFINITY, ymn = -INFINITY, gmx = -INFINITY, gmn = -INFINITY;
    int gimx = 1 << 30, gimn = 1 << 30, d0 = 0, d1 = 0;
    for (int i = lane; i < k; i += 64) {
      const float yv = yr[i], gv = gr[i];
      if (yv > ymx) ymx = yv;
      if (-yv > ymn) ymn = -yv;
      if (gv > gmx) { gmx = gv; gimx = i; }
      if (-gv > gmn) { gmn = -gv; gimn = i; }
    }
    wave_argmax(ymx, d0);
    wave_argmax(ymn, d1);
    wave_argmax(gmx, gimx);
    wave_argmax(gmn, gimn);
    if (ENV) {
      if (!BWD) {
        if (lane == 0) total += fabsf(ymx - gmx) + fabsf(ymn - gmn);
      } else {
        // d/dg |ymx - max(g)| = -sign(ymx - gmx) at argmax(g);  d/dg |ymn - max(-g)| = +sign(ymn - gmn) at argmax(-g)
        const float k0 = scale * (gscale ? *gscale : 1.f);
        const float c1 = -((ymx > gmx) ? 1.f : ((ymx < gmx) ? -1.f : 0.f)) * k0;
        const float c2 = ((ymn > gmn) ? 1.f : ((ymn < gmn) ? -1.f : 0.f)) * k0;
        float* dr = dg + (size_t)r * L + w0;
        for (int i = lane; i < k; i += 64) {
          float v = 0.f;
          if (i == gimx) v += c1;
          if (i == gimn) v += c2;
          dr[i] = v;
        }
      }
      continue;
    }
    const float dy_ = fabsf(ymx + ymn), sg = gmx + gmn, dgv = fabsf(sg);
    if (!BWD) {
      if (lane == 0) total += fabsf(dy_ - dgv);
    } else {
      // d/dg of |dy_ - |sg||: -sign(dy_ - dgv) * sign(sg) flows +1 to argmax(g) and -1 to argmax(-g)
      const float s1 = (dy_ > dgv) ? 1.f : ((dy_ < dgv) ? -1.f : 0.f);
      const float s2 = (sg > 0.f) ? 1.f : ((sg < 0.f) ? -1.f : 0.f);
      const float c = -s1 * s2 * scale * (gscale ? *gscale : 1.f);
      float* dr = dg + (size_t)r * L + w0;
      for (int i = lane; i < k; i += 64) {
        float v = 0.f;
        if (i == gimx) v += c;
        if (i == gimn) v -= c;
        dr[i] = v;
      }
    }
  }
  if (!BWD) {
    __shared__ float red[4];
    total = rtg_block_sum(total, red);
    if (threadIdx.x == 0) ws[blockIdx.x] = total * scale;
  } else {
    // tail samples beyond the last full window get no gradient
    const int tail0 = W * k;
    for (long long i = (long long)blockIdx.x * RTG_THREADS + threadIdx.x; i < (long long)rows * (L - tail0);
         i += (long long)gridDim.x * RTG_THREADS) {
      const int r = (int)(i / (L - tail0)), o = (int)(i - (long long)r * (L - tail0));
      dg[(size_t)r * L + tail0 + o] = 0.f;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// AdamW over a flat buffer
// ---------------------------------------------------------------------------------------------------------------
// torch.optim.AdamW, single-tensor form, in torch's own order of operations (torch/optim/adamw.py / adam.py
// _single_tensor_adam): scalar factors (bias corrections, step size, decay factor) in double like the Python side
// computes them, element math in fp32: p *= 1 - lr*wd; m = lerp(m, g, 1-b1); v = v*b2 + (1-b2)*g*g;
// p -= step_size * m / (sqrt(v) / sqrt(bc2) + eps).
__global__ __launch_bounds__(RTG_THREADS) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                            float* __restrict__ m, float* __restrict__ v, long long n,
                                                            const float* __restrict__ step_state,
                                                            const float* __restrict__ loss_flag, double lr, double b1,
                                                            double b2, double eps_d, double wd, float gscale) {
  if (loss_flag && isnan(*loss_flag)) return;
  const double t = (double)step_state[0] + 1.0;
  const double bc1 = 1.0 - pow(b1, t), bc2 = 1.0 - pow(b2, t);
  const float step_size = (float)(lr / bc1), bc2s = (float)sqrt(bc2), decay = (float)(1.0 - lr * wd);
  const float w1 = (float)(1.0 - b1), fb2 = (float)b2, w2 = (float)(1.0 - b2), eps = (float)eps_d;
  for (long long i = (long long)blockIdx.x * RTG_THREADS + threadIdx.x; i < n; i += (long long)gridDim.x * RTG_THREADS) {
    const float gi = g[i] * gscale;
    const float m0 = m[i];
    const float mi = m0 + w1 * (gi - m0);
    const float vi = v[i] * fb2 + w2 * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] = p[i] * decay - step_size * mi / (sqrtf(vi) / bc2s + eps);
  }
}

__global__ void adamw_bump_kernel(float* step_state, const float* loss_flag) {
  if (loss_flag && isnan(*loss_flag)) return;
  step_state[0] += 1.f;
}

}  // namespace

#define RTG_REQ(c) \
  if (!(c)) return RTG_ENULL
// 16-byte accesses: n a multiple of 4 and every (non-null) pointer 16-byte aligned
static inline bool vec4_ok(long long n, const void* a, const void* b, const void* c = nullptr, const void* d = nullptr) {
  const uintptr_t bits = reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c) |
                         reinterpret_cast<uintptr_t>(d);
  return (n & 3) == 0 && (bits & 15) == 0;
}
#define RTG_LAUNCH(k, g, b, sh, st, ...)                        \
  RTG_KLAUNCH(k, dim3(g), dim3(b), sh, (hipStream_t)st, __VA_ARGS__); \
  return rtg_launch_status()

extern "C" int rtg_noise_lrelu_fwd(const float* x, const float* w, const float* u_in, float* out, long long n,
                                   float slope, unsigned long long seed, const float* salt_dev, void* stream) {
  RTG_REQ(x && w && out);
  if (n < 1) return RTG_EINVAL;
  if (vec4_ok(n, x, out, u_in)) {
    RTG_LAUNCH(noise_fwd_kernel<4>, grid_for(n / 4, 2), RTG_THREADS, 0, stream, x, w, u_in, out, n, slope, seed, salt_dev);
  }
  RTG_LAUNCH(noise_fwd_kernel<1>, grid_for(n), RTG_THREADS, 0, stream, x, w, u_in, out, n, slope, seed, salt_dev);
}

extern "C" int rtg_noise_lrelu_bwd(const float* x, const float* w, const float* u_in, const float* dy, float* dx,
                                   float* dw_part, int n_blocks, long long n, float slope, unsigned long long seed,
                                   const float* salt_dev, void* stream) {
  RTG_REQ(x && w && dy && dx && dw_part);
  if (n < 1 || n_blocks < 1 || n_blocks > MAX_GRID) return RTG_EINVAL;
  if (vec4_ok(n, x, dy, dx, u_in)) {
    RTG_LAUNCH(noise_bwd_kernel<4>, n_blocks, RTG_THREADS, 0, stream, x, w, u_in, dy, dx, dw_part, n, slope, seed, salt_dev);
  }
  RTG_LAUNCH(noise_bwd_kernel<1>, n_blocks, RTG_THREADS, 0, stream, x, w, u_in, dy, dx, dw_part, n, slope, seed, salt_dev);
}

extern "C" int rtg_noise_lrelu_bwd_acc(const float* x, const float* w, const float* u_in, const float* dy, float* dx,
                                       float* dw_part, int n_blocks, long long n, float slope, unsigned long long seed,
                                       const float* salt_dev, float* dw_acc, void* stream) {
  RTG_REQ(dw_acc);
  const int rc = rtg_noise_lrelu_bwd(x, w, u_in, dy, dx, dw_part, n_blocks, n, slope, seed, salt_dev, stream);
  if (rc != RTG_OK) return rc;
  RTG_LAUNCH(reduce_acc_kernel, 1, RTG_THREADS, 0, stream, (const float*)dw_part, n_blocks, dw_acc);
}

extern "C" int rtg_axpby(const float* a, const float* b, float* out, long long n, float alpha, float beta,
                         int accumulate, void* stream) {
  RTG_REQ(a && out);
  if (n < 1) return RTG_EINVAL;
  if (vec4_ok(n, a, out, b)) {
    RTG_LAUNCH(axpby_kernel<4>, grid_for(n / 4, 2), RTG_THREADS, 0, stream, a, b, out, n, alpha, beta, accumulate);
  }
  RTG_LAUNCH(axpby_kernel<1>, grid_for(n), RTG_THREADS, 0, stream, a, b, out, n, alpha, beta, accumulate);
}

extern "C" int rtg_channel_sum(const float* x, float* out, int B, int C, int L, float* ws, void* stream) {
  RTG_REQ(x && out && ws);
  if (B < 1 || C < 1 || L < 1 || C > 65535) return RTG_EINVAL;
  RTG_KLAUNCH(channel_sum_stage1, dim3(C, CSUM_SLICES), dim3(RTG_THREADS), 0, (hipStream_t)stream, x, ws, B, C, L);
  const int st = rtg_launch_status();
  if (st != RTG_OK) return st;
  RTG_LAUNCH(channel_sum_stage2, rtg_ceil_div(C, RTG_THREADS), RTG_THREADS, 0, stream, ws, out, C);
}

extern "C" int rtg_lrelu_bwd(const float* dy, const float* ref, float* dx, long long n, float slope, void* stream) {
  RTG_REQ(dy && ref && dx);
  if (n < 1) return RTG_EINVAL;
  if (vec4_ok(n, dy, ref, dx)) {
    RTG_LAUNCH(lrelu_bwd_kernel<4>, grid_for(n / 4, 2), RTG_THREADS, 0, stream, dy, ref, dx, n, slope);
  }
  RTG_LAUNCH(lrelu_bwd_kernel<1>, grid_for(n), RTG_THREADS, 0, stream, dy, ref, dx, n, slope);
}

extern "C" int rtg_avgpool4s2_fwd(const float* x, float* out, int rows, int L, void* stream) {
  RTG_REQ(x && out);
  if (rows < 1 || L < 4 || (L & 1)) return RTG_EINVAL;
  RTG_LAUNCH(avgpool_fwd_kernel, grid_for((long long)rows * (L / 2)), RTG_THREADS, 0, stream, x, out, rows, L);
}

extern "C" int rtg_bf16_encode(const float* src, void* dst_bf16, long long n, float slope, void* stream) {
  RTG_REQ(src && dst_bf16);
  if (n < 1 || !(slope > 0.f)) return RTG_EINVAL;
  const int vec = ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst_bf16)) & 15) == 0 ? 1 : 0;
  RTG_LAUNCH(bf16_encode_kernel, grid_for(n / 8 + 1, 2), RTG_THREADS, 0, stream, src, (unsigned short*)dst_bf16, n, slope, vec);
}

extern "C" int rtg_bf16_decode(const void* src_bf16, float* dst, long long n, float slope, void* stream) {
  RTG_REQ(src_bf16 && dst);
  if (n < 1 || !(slope > 0.f)) return RTG_EINVAL;
  const int vec = ((reinterpret_cast<uintptr_t>(src_bf16) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0 ? 1 : 0;
  RTG_LAUNCH(bf16_decode_kernel, grid_for(n / 8 + 1, 2), RTG_THREADS, 0, stream, (const unsigned short*)src_bf16, dst, n,
             1.f / slope, vec);
}

extern "C" int rtg_avgpool4s2_bwd(const float* dy, float* dx, int rows, int L, void* stream) {
  RTG_REQ(dy && dx);
  if (rows < 1 || L < 4 || (L & 1)) return RTG_EINVAL;
  RTG_LAUNCH(avgpool_bwd_kernel, grid_for((long long)rows * L), RTG_THREADS, 0, stream, dy, dx, rows, L);
}

extern "C" int rtg_period_fold_fwd(const float* y, float* out, int B, int T, int p, int H, void* stream) {
  RTG_REQ(y && out);
  if (B < 1 || T < 2 || p < 1 || H != (T + p - 1) / p || H * p - T >= T) return RTG_EINVAL;
  RTG_LAUNCH(fold_fwd_kernel, grid_for((long long)B * p * H), RTG_THREADS, 0, stream, y, out, B, T, p, H);
}

extern "C" int rtg_period_fold_bwd(const float* dout, float* dy, int B, int T, int p, int H, void* stream) {
  RTG_REQ(dout && dy);
  if (B < 1 || T < 2 || p < 1 || H != (T + p - 1) / p || H * p - T >= T) return RTG_EINVAL;
  RTG_LAUNCH(fold_bwd_kernel, grid_for((long long)B * T), RTG_THREADS, 0, stream, dout, dy, B, T, p, H);
}

static int fill_jobs(LossJobs* lj, int kind, const RtgLossJob* jobs, int n_jobs, bool bwd) {
  if (!jobs) return RTG_ENULL;
  if (n_jobs < 1 || n_jobs > RTG_MAX_LOSS_JOBS) return RTG_EINVAL;
  if (kind != RTG_LOSS_L1 && kind != RTG_LOSS_L1_L1LOG && kind != RTG_LOSS_MSE_TARGET && kind != RTG_LOSS_MSE_REL &&
      kind != RTG_LOSS_L1_ENC)
    return RTG_EINVAL;
  lj->n_jobs = n_jobs;
  for (int i = 0; i < n_jobs; ++i) {
    if (!jobs[i].a || jobs[i].n < 1) return RTG_EINVAL;
    if (kind != RTG_LOSS_MSE_TARGET && !jobs[i].b) return RTG_ENULL;
    if (bwd && !jobs[i].da && !jobs[i].db) return RTG_ENULL;
    lj->job[i] = jobs[i];
  }
  return RTG_OK;
}

extern "C" int rtg_loss_fwd(int kind, const RtgLossJob* jobs, int n_jobs, float* ws, float* loss_out, void* stream) {
  RTG_REQ(ws && loss_out);
  LossJobs lj;
  int st = fill_jobs(&lj, kind, jobs, n_jobs, false);
  if (st) return st;
  RTG_KLAUNCH(loss_fwd_kernel, dim3(LOSS_GX, n_jobs), dim3(RTG_THREADS), 0, (hipStream_t)stream, kind, lj, ws);
  st = rtg_launch_status();
  if (st) return st;
  RTG_LAUNCH(loss_finish_kernel, 1, 64, 0, stream, n_jobs * LOSS_GX, ws, loss_out);
}

extern "C" int rtg_loss_bwd(int kind, const RtgLossJob* jobs, int n_jobs, const float* gscale, void* stream) {
  LossJobs lj;
  int st = fill_jobs(&lj, kind, jobs, n_jobs, true);
  if (st) return st;
  long long mx = 0;
  for (int i = 0; i < n_jobs; ++i) mx = jobs[i].n > mx ? jobs[i].n : mx;
  int gx = grid_for(mx);
  if (gx > 512) gx = 512;
  RTG_LAUNCH(loss_bwd_kernel, dim3(gx, n_jobs), RTG_THREADS, 0, stream, kind, lj, gscale);
}

template <bool ENV>
static int dyn_fwd_impl(const float* y, const float* g, int rows, int L, int k, float w, float* ws, float* loss_out,
                        void* stream) {
  RTG_REQ(y && g && ws && loss_out);
  if (rows < 1 || k < 1 || L < k) return RTG_EINVAL;
  const int nwin = rows * (L / k);
  int gx = (nwin + 3) / 4;
  if (gx > 256) gx = 256;
  RTG_KLAUNCH((dyn_kernel<false, ENV>), dim3(gx), dim3(RTG_THREADS), 0, (hipStream_t)stream, y, g, rows, L, k,
                     w / (float)nwin, (const float*)nullptr, ws, (float*)nullptr);
  int st = rtg_launch_status();
  if (st) return st;
  RTG_LAUNCH(loss_finish_kernel, 1, 64, 0, stream, gx, ws, loss_out);
}

template <bool ENV>
static int dyn_bwd_impl(const float* y, const float* g, int rows, int L, int k, float w, const float* gscale, float* dg,
                        void* stream) {
  RTG_REQ(y && g && dg);
  if (rows < 1 || k < 1 || L < k) return RTG_EINVAL;
  const int nwin = rows * (L / k);
  int gx = (nwin + 3) / 4;
  if (gx > 256) gx = 256;
  RTG_LAUNCH((dyn_kernel<true, ENV>), gx, RTG_THREADS, 0, stream, y, g, rows, L, k, w / (float)nwin, gscale,
             (float*)nullptr, dg);
}

extern "C" int rtg_dyn_loss_fwd(const float* y, const float* g, int rows, int L, int k, float w, float* ws,
                                float* loss_out, void* stream) {
  return dyn_fwd_impl<false>(y, g, rows, L, k, w, ws, loss_out, stream);
}
extern "C" int rtg_dyn_loss_bwd(const float* y, const float* g, int rows, int L, int k, float w, const float* gscale,
                                float* dg, void* stream) {
  return dyn_bwd_impl<false>(y, g, rows, L, k, w, gscale, dg, stream);
}
extern "C" int rtg_env_loss_fwd(const float* y, const float* g, int rows, int L, int k, float w, float* ws,
                                float* loss_out, void* stream) {
  return dyn_fwd_impl<true>(y, g, rows, L, k, w, ws, loss_out, stream);
}
extern "C" int rtg_env_loss_bwd(const float* y, const float* g, int rows, int L, int k, float w, const float* gscale,
                                float* dg, void* stream) {
  return dyn_bwd_impl<true>(y, g, rows, L, k, w, gscale, dg, stream);
}

// ---------------------------------------------------------------------------------------------------------------
// strip_mirror_loss (loss.py:86-98): u_i = y[2i] - y[2i+1], d_i = u_i - mean(u), f(d) = -log(min(|d| + 1e-9, 1))
// ---------------------------------------------------------------------------------------------------------------
constexpr int SM_GX = 128;
__device__ __forceinline__ float sm_u(const float* __restrict__ y, int L, int half, long long i) {
  const long long r = i / half, c = i - r * half;
  const float* p = y + r * L + 2 * c;
  return p[0] - p[1];
}
// PASS 0: partial sums of u.  PASS 1: partial sums of f(d) and f'(d), with mean(u) = stats[0] / n
template <int PASS>
__global__ __launch_bounds__(RTG_THREADS) void sm_reduce_kernel(const float* __restrict__ y, int L, int half,
                                                                long long n, const float* __restrict__ stats,
                                                                float* __restrict__ ws) {
  __shared__ float red[4];
  const float mu = PASS ? stats[0] / (float)n : 0.f;
  float s0 = 0.f, s1 = 0.f;
  for (long long i = (long long)blockIdx.x * RTG_THREADS + threadIdx.x; i < n; i += (long long)SM_GX * RTG_THREADS) {
    const float u = sm_u(y, L, half, i);
    if (PASS == 0) {
      s0 += u;
    } else {
      const float d = u - mu, m = fabsf(d) + 1e-9f;
      if (m <= 1.f) {                       // clamp_max passes the gradient up to and including the bound
        s0 += -logf(m);
        s1 += -((d > 0.f) ? 1.f : ((d < 0.f) ? -1.f : 0.f)) / m;
      }
    }
  }
  s0 = rtg_block_sum(s0, red);
  s1 = rtg_block_sum(s1, red);
  if (threadIdx.x == 0) {
    ws[blockIdx.x] = s0;
    ws[SM_GX + blockIdx.x] = s1;
  }
}
// sums the partials in fixed order: PASS 0 -> stats[0] = sum u; PASS 1 -> stats[1] = sum f', *loss_out += w * sum f / n
__global__ __launch_bounds__(64) void sm_finish_kernel(int pass, const float* __restrict__ ws, float* __restrict__ stats,
                                                       float* __restrict__ loss_out, float w_over_n) {
  float a = 0.f, b = 0.f;
  for (int i = threadIdx.x; i < SM_GX; i += 64) { a += ws[i]; b += ws[SM_GX + i]; }
  a = rtg_wave_sum(a);
  b = rtg_wave_sum(b);
  if (threadIdx.x == 0) {
    if (pass == 0) {
      stats[0] = a;
    } else {
      stats[1] = b;
      *loss_out += a * w_over_n;
    }
  }
}
__global__ __launch_bounds__(RTG_THREADS) void sm_bwd_kernel(const float* __restrict__ y, int rows, int L, int half,
                                                             long long n, const float* __restrict__ stats, float k,
                                                             const float* __restrict__ gscale, float* __restrict__ dy) {
  const float mu = stats[0] / (float)n, mfp = stats[1] / (float)n, kk = k * (gscale ? *gscale : 1.f);
  for (long long i = (long long)blockIdx.x * RTG_THREADS + threadIdx.x; i < n; i += (long long)gridDim.x * RTG_THREADS) {
    const long long r = i / half, c = i - r * half;
    const float* p = y + r * L + 2 * c;
    const float d = (p[0] - p[1]) - mu, m = fabsf(d) + 1e-9f;
    const float fp = (m <= 1.f) ? -((d > 0.f) ? 1.f : ((d < 0.f) ? -1.f : 0.f)) / m : 0.f;
    const float gu = kk * (fp - mfp);       // d loss / d u_i: the mean(u) term couples every element
    float* q = dy + r * L + 2 * c;
    q[0] = gu;
    q[1] = -gu;
  }
  if ((L & 1) && blockIdx.x == 0)
    for (int r = threadIdx.x; r < rows; r += RTG_THREADS) dy[(size_t)r * L + L - 1] = 0.f;
}

extern "C" int rtg_strip_mirror_fwd(const float* y, int rows, int L, float w, float* ws, float* stats, float* loss_out,
                                    void* stream) {
  RTG_REQ(y && ws && stats && loss_out);
  if (rows < 1 || L < 2) return RTG_EINVAL;
  const int half = L / 2;
  const long long n = (long long)rows * half;
  hipStream_t s = (hipStream_t)stream;
  RTG_KLAUNCH(sm_reduce_kernel<0>, dim3(SM_GX), dim3(RTG_THREADS), 0, s, y, L, half, n, (const float*)stats, ws);
  RTG_KLAUNCH(sm_finish_kernel, dim3(1), dim3(64), 0, s, 0, (const float*)ws, stats, loss_out, 0.f);
  RTG_KLAUNCH(sm_reduce_kernel<1>, dim3(SM_GX), dim3(RTG_THREADS), 0, s, y, L, half, n, (const float*)stats, ws);
  RTG_LAUNCH(sm_finish_kernel, 1, 64, 0, stream, 1, (const float*)ws, stats, loss_out, w / (float)n);
}

extern "C" int rtg_strip_mirror_bwd(const float* y, int rows, int L, float w, const float* stats, const float* gscale,
                                    float* dy, void* stream) {
  RTG_REQ(y && stats && dy);
  if (rows < 1 || L < 2) return RTG_EINVAL;
  const int half = L / 2;
  const long long n = (long long)rows * half;
  RTG_LAUNCH(sm_bwd_kernel, grid_for(n), RTG_THREADS, 0, stream, y, rows, L, half, n, stats, w / (float)n, gscale, dy);
}

extern "C" int rtg_adamw(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long long n,
                         float* step_state, const float* loss_flag, double lr, double beta1, double beta2, double eps,
                         double weight_decay, float grad_scale, void* stream) {
  RTG_REQ(params && grads && exp_avg && exp_avg_sq && step_state);
  if (n < 1) return RTG_EINVAL;
  RTG_KLAUNCH(adamw_kernel, dim3(grid_for(n)), dim3(RTG_THREADS), 0, (hipStream_t)stream, params, grads, exp_avg,
                     exp_avg_sq, n, step_state, loss_flag, lr, beta1, beta2, eps, weight_decay, grad_scale);
  int st = rtg_launch_status();
  if (st) return st;
  RTG_LAUNCH(adamw_bump_kernel, 1, 1, 0, stream, step_state, loss_flag);
}

extern "C" int rtg_stream_create(int priority, void** stream) {
  if (!stream) return RTG_ENULL;
  int lo = 0, hi = 0;                                 // (numerically: hi <= priority <= lo, hi = highest priority)
  if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) return RTG_EINVAL;
  int p = priority < hi ? hi : (priority > lo ? lo : priority);
  hipStream_t s = nullptr;
  if (hipStreamCreateWithPriority(&s, hipStreamNonBlocking, p) != hipSuccess) return RTG_EINVAL;
  *stream = (void*)s;
  return 0;
}

extern "C" int rtg_stream_destroy(void* stream) {
  if (!stream) return RTG_ENULL;
  return hipStreamDestroy((hipStream_t)stream) == hipSuccess ? 0 : RTG_EINVAL;
}

extern "C" int rtg_stream_end_capture(void* stream) {
  if (!stream) return RTG_ENULL;
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing((hipStream_t)stream, &st) != hipSuccess) { (void)hipGetLastError(); return RTG_EINVAL; }
  if (st == hipStreamCaptureStatusNone) return 0;
  hipGraph_t g = nullptr;
  (void)hipStreamEndCapture((hipStream_t)stream, &g);          // (an invalidated capture reports an error and still ends)
  if (g) (void)hipGraphDestroy(g);
  (void)hipGetLastError();
  return 1;
}

// ---- weighted sum of device scalars (the step's loss total) and its backward fan-out: one launch each instead of an ATen
// mul / add per term (train.py:137-158, 170-189 of the reference add the terms one by one)
__global__ void scalar_wsum_kernel(const RtgScalarTerms t, float* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float s = 0.f;
  for (int i = 0; i < t.n; ++i) s += t.w[i] * *t.p[i];          // terms in list order, one rounding per product and sum
  out[0] = s;
}
__global__ void scalar_fanout_kernel(const RtgScalarTerms t, const float* __restrict__ g, float* __restrict__ out) {
  const int i = threadIdx.x;
  if (blockIdx.x == 0 && i < t.n) out[i] = t.w[i] * g[0];
}

extern "C" int rtg_scalar_wsum(const RtgScalarTerms* terms, float* out, void* stream) {
  RTG_REQ(terms && out);
  if (terms->n < 1 || terms->n > RTG_MAX_SCALAR_TERMS) return RTG_EINVAL;
  for (int i = 0; i < terms->n; ++i)
    if (!terms->p[i]) return RTG_ENULL;
  RTG_LAUNCH(scalar_wsum_kernel, 1, 64, 0, stream, *terms, out);
}

extern "C" int rtg_scalar_fanout(const RtgScalarTerms* terms, const float* g, float* out, void* stream) {
  RTG_REQ(terms && g && out);
  if (terms->n < 1 || terms->n > RTG_MAX_SCALAR_TERMS) return RTG_EINVAL;
  RTG_LAUNCH(scalar_fanout_kernel, 1, 64, 0, stream, *terms, g, out);
}

extern "C" int rtg_abi_version(void) { return RTG_ABI_VERSION; }
// (a library compiled with an ablation / diagnostic define says so: rtg/lib.py refuses it as the product library)
// RTG_ABLATION: set by every dev build script (tools/dev_build.sh, tools/dbg/abl.sh) whatever -DRTG_EXP_* / -DRTG_STAMPS
// flag it compiles into whichever source — those scripts always recompile THIS file with it
#if defined(RTG_ABLATION) || defined(RTG_EXP_NOMFMA_BF) || defined(RTG_EXP_EMPTY) || defined(RTG_EXP_SKIPLOOP) || \
    defined(RTG_EXP_NOGENERAL) || defined(RTG_STAMPS)
#define RTG_BUILD_KIND " ABLATION"
#else
#define RTG_BUILD_KIND ""
#endif
extern "C" const char* rtg_build_info(void) { return "librtg gfx950 fp32-mfma" RTG_BUILD_KIND " " __DATE__ " " __TIME__; }
