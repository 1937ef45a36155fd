// rtg_wgrad_thin.hip — weight / bias gradients of the two degenerate conv shapes as bandwidth kernels.
//
//   one input channel   conv_pre (generator.py:682), the first conv of every discriminator (discrminator.py:38,157):
//                       dW[m][j] = sum_{b,t} gy[b,m,t] * x[b,0,t*s - p + j*d]   — reads gy once (the big operand)
//   one output channel  conv_post of G and of every discriminator (generator.py:722, discrminator.py:45,163):
//                       dW[c][j] = sum_{b,t} gy[b,0,t] * pre(x[b,c,t - p + j*d]) — reads x once
// On the matrix cores (rtg_wgrad_kernel.h) these fill 1/16 .. 1/32 of a tile and run at 1 .. 9 TFLOP/s for 30 .. 60 us a
// launch; the 30 of them in a train step cost ~1.2 ms.  Here every element of the big operand is read exactly once by a
// coalesced load, the small operand sits in LDS, each thread keeps its rows' accumulators in registers over all the
// tiles of its block and the cross-lane reduction happens once per block.  Output: one split partial per block in the
// layout rtg_weightnorm_backward reduces (fixed order: bitwise reproducible).  Exposed as block-shape code 7 of
// RtgWgradDesc.shape_cfg.
#include "rtg_common.h"

namespace {

struct WtArgs {
  const float *x, *dy, *aux;
  float* part;
  long long part_stride;
  int B, C, L_in, M, K, stride, dil, pad, Q, dy_L;
  int pre_mode, gy_mode;
  float pre_slope, gy_slope, gy_scale;
  int n_t, items, splits;
};

__device__ __forceinline__ float gy_eff(float g, float av, int mode, float slope, float scale) {
  if (mode == RTG_PRE_MUL_DLRELU) g *= (av > 0.f ? 1.f : slope);
  else if (mode == RTG_PRE_MUL_DTANH) g *= fmaf(-av, av, 1.f);
  return g * scale;
}

constexpr int kPT = 4 * RTG_THREADS;          // positions per tile; a wave owns its rows over ALL of them: lane + 64 * i

// ---- one input channel.  Wave w owns rows w, w + 4, ... (RW of them): every gy element is read once, by the wave that
// owns its row, lane-consecutive along the positions; the x window of the tile is shared through LDS.
template <int KT, int RW>
__global__ __launch_bounds__(RTG_THREADS) void wgrad_cin1_kernel(const WtArgs a) {
  extern __shared__ __attribute__((aligned(16))) float xs[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float acc[RW][KT + 1];
#pragma unroll
  for (int r = 0; r < RW; ++r)
#pragma unroll
    for (int j = 0; j <= KT; ++j) acc[r][j] = 0.f;
  const int per = (a.items + a.splits - 1) / a.splits;
  const int lo = blockIdx.x * per, hi = min(a.items, lo + per);
  const int W = (kPT - 1) * a.stride + (KT - 1) * a.dil + 1;
  for (int item = lo; item < hi; ++item) {
    const int b = item / a.n_t, t0 = (item - b * a.n_t) * kPT;
    const float* xr = a.x + (size_t)b * a.L_in;
    const int g0 = t0 * a.stride - a.pad;
    __syncthreads();
    for (int i = tid; i < W; i += RTG_THREADS) {
      const int pos = g0 + i;
      float v = (pos >= 0 && pos < a.L_in) ? xr[pos] : 0.f;
      if (a.pre_mode == RTG_PRE_LRELU) v = v > 0.f ? v : v * a.pre_slope;
      xs[i] = v;
    }
    __syncthreads();
    // (measured: requesting step i + 1's gy values ahead by hand is SLOWER here — 35 vs 22 us on the MPD first layers:
    // the accumulators already take 128 registers — so the plain loop stays; the compiler batches the unrolled loads)
#pragma unroll 4
    for (int i = 0; i < kPT / 64; ++i) {
      const int tl = lane + 64 * i, t = t0 + tl;
      if (t >= a.Q) continue;
      float xv[KT];
#pragma unroll
      for (int j = 0; j < KT; ++j) xv[j] = xs[tl * a.stride + j * a.dil];
      float g[RW];
#pragma unroll
      for (int r = 0; r < RW; ++r) {
        const int m = wave + 4 * r;
        const size_t o = ((size_t)b * a.M + m) * a.dy_L + t;
        g[r] = m < a.M ? a.dy[o] : 0.f;
        if (a.gy_mode != RTG_PRE_NONE && m < a.M) g[r] = gy_eff(g[r], a.aux[o], a.gy_mode, a.gy_slope, 1.f);
      }
#pragma unroll
      for (int r = 0; r < RW; ++r) {
        const float gs = g[r] * a.gy_scale;
#pragma unroll
        for (int j = 0; j < KT; ++j) acc[r][j] = fmaf(gs, xv[j], acc[r][j]);
        acc[r][KT] += gs;
      }
    }
  }
  // one partial per block: [M][K] weights, then [M] biases
  float* p = a.part + (size_t)blockIdx.x * a.part_stride;
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    const int m = wave + 4 * r;
#pragma unroll
    for (int j = 0; j <= KT; ++j) {
      const float s = rtg_wave_sum(acc[r][j]);
      if (lane == 0 && m < a.M) {
        if (j < KT) p[m * KT + j] = s;
        else p[a.M * KT + m] = s;
      }
    }
  }
}

// ---- one output channel, long rows.  Wave w owns channels w, w + 4, ... (RW of them); the gy window of the tile sits in
// LDS (activation derivative and scale applied at staging).  x position u meets tap j at output t = u + pad - j * dil.
template <int KT, int RW>
__global__ __launch_bounds__(RTG_THREADS) void wgrad_cout1_long_kernel(const WtArgs a) {
  extern __shared__ __attribute__((aligned(16))) float gs[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float acc[RW][KT];
#pragma unroll
  for (int r = 0; r < RW; ++r)
#pragma unroll
    for (int j = 0; j < KT; ++j) acc[r][j] = 0.f;
  float bacc = 0.f;
  const int per = (a.items + a.splits - 1) / a.splits;
  const int lo = blockIdx.x * per, hi = min(a.items, lo + per);
  const int halo = (KT - 1) * a.dil;
  const int W = kPT + halo;
  for (int item = lo; item < hi; ++item) {
    const int b = item / a.n_t, u0 = (item - b * a.n_t) * kPT;
    const int tw0 = u0 + a.pad - halo;                       // output position of LDS column 0
    __syncthreads();
    for (int i = tid; i < W; i += RTG_THREADS) {
      const int t = tw0 + i;
      float g = 0.f;
      if (t >= 0 && t < a.Q) {
        const size_t o = (size_t)b * a.dy_L + t;
        g = gy_eff(a.dy[o], a.gy_mode != RTG_PRE_NONE ? a.aux[o] : 0.f, a.gy_mode, a.gy_slope, a.gy_scale);
      }
      gs[i] = g;
    }
    __syncthreads();
    auto xload = [&](int i, float (&xv)[RW]) __attribute__((always_inline)) {
      const int u = u0 + lane + 64 * i;
#pragma unroll
      for (int r = 0; r < RW; ++r) {
        const int c = wave + 4 * r;
        const bool ok = c < a.C && u < a.L_in;
        float v = ok ? a.x[((size_t)b * a.C + c) * a.L_in + u] : 0.f;
        if (a.pre_mode == RTG_PRE_LRELU) v = v > 0.f ? v : v * a.pre_slope;
        xv[r] = v;
      }
    };
    auto step = [&](int i, const float (&xv)[RW]) __attribute__((always_inline)) {
      const int ul = lane + 64 * i;
      float gv[KT];
#pragma unroll
      for (int j = 0; j < KT; ++j) gv[j] = gs[ul + halo - j * a.dil];          // t = u + pad - j * dil
#pragma unroll
      for (int r = 0; r < RW; ++r)
#pragma unroll
        for (int j = 0; j < KT; ++j) acc[r][j] = fmaf(xv[r], gv[j], acc[r][j]);
    };
    float xa[RW], xb[RW];
    xload(0, xa);
    for (int i = 0; i < kPT / 64; i += 2) {
      xload(i + 1, xb);
      step(i, xa);
      if (i + 2 < kPT / 64) xload(i + 2, xa);
      step(i + 1, xb);
    }
    // bias gradient: the tile's own outputs [u0, u0 + kPT) (LDS column of output t is t - tw0)
    if (wave == 0) {
#pragma unroll
      for (int i = 0; i < 4 * RTG_THREADS / 64; ++i) {
        const int t = u0 + lane + 64 * i;
        if (t < a.Q && t < u0 + kPT) bacc += gs[t - tw0];
      }
    }
  }
  float* p = a.part + (size_t)blockIdx.x * a.part_stride;
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    const int c = wave + 4 * r;
#pragma unroll
    for (int j = 0; j < KT; ++j) {
      const float s = rtg_wave_sum(acc[r][j]);
      if (lane == 0 && c < a.C) p[c * KT + j] = s;
    }
  }
  if (wave == 0) {
    const float s = rtg_wave_sum(bacc);
    if (lane == 0) p[a.C * KT] = s;
  }
}

// ---- one output channel, short rows (Q <= 256: the discriminator conv_posts, 512 channels x 10 .. 128 positions):
// thread = channel; the clip's gy row sits in LDS; a thread walks its channel's row (lanes stride Q floats apart: every
// cache line of the contiguous [256 channels][Q] slab is used by consecutive iterations of the same wave).
template <int KT>
__global__ __launch_bounds__(RTG_THREADS) void wgrad_cout1_short_kernel(const WtArgs a) {
  __shared__ float gs[256 + 2 * 16 * 8];
  const int tid = threadIdx.x;
  const int c = blockIdx.y * RTG_THREADS + tid;
  float acc[KT];
#pragma unroll
  for (int j = 0; j < KT; ++j) acc[j] = 0.f;
  float bacc = 0.f;
  const int per = (a.B + a.splits - 1) / a.splits;
  const int lo = blockIdx.x * per, hi = min(a.B, lo + per);
  const int halo = (KT - 1) * a.dil;
  for (int b = lo; b < hi; ++b) {
    __syncthreads();
    // LDS column i holds output t = i - halo (zeros outside [0, Q))
    for (int i = tid; i < a.Q + 2 * halo; i += RTG_THREADS) {
      const int t = i - halo;
      float g = 0.f;
      if (t >= 0 && t < a.Q) {
        const size_t o = (size_t)b * a.dy_L + t;
        g = gy_eff(a.dy[o], a.gy_mode != RTG_PRE_NONE ? a.aux[o] : 0.f, a.gy_mode, a.gy_slope, a.gy_scale);
      }
      gs[i] = g;
    }
    __syncthreads();
    if (tid == 0 && blockIdx.y == 0)
      for (int t = 0; t < a.Q; ++t) bacc += gs[t + halo];
    if (c < a.C) {
      const float* xr = a.x + ((size_t)b * a.C + c) * a.L_in;
      for (int u0 = 0; u0 < a.L_in; u0 += 8) {
        float xv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) xv[e] = u0 + e < a.L_in ? xr[u0 + e] : 0.f;       // 8 loads in flight per thread
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float v = xv[e];
          if (a.pre_mode == RTG_PRE_LRELU) v = v > 0.f ? v : v * a.pre_slope;
          const int u = u0 + e < a.L_in ? u0 + e : 0;                                // (v == 0 past the row)
          const float* gp = gs + u + a.pad + halo;             // output t = u + pad - j * dil
#pragma unroll
          for (int j = 0; j < KT; ++j) acc[j] = fmaf(v, gp[-j * a.dil], acc[j]);
        }
      }
    }
  }
  float* p = a.part + (size_t)blockIdx.x * a.part_stride;
  if (c < a.C) {
#pragma unroll
    for (int j = 0; j < KT; ++j) p[c * KT + j] = acc[j];
  }
  if (tid == 0 && blockIdx.y == 0) p[a.C * KT] = bacc;
}

}  // namespace

// rtg_thin2d.hip: the first Conv2d of StftDiscriminator (2 input channels)
bool rtg_thin2d_wgrad_ok(const RtgWgradDesc* d);
int rtg_thin2d_wgrad_splits(const RtgWgradDesc* d);
int rtg_thin2d_wgrad_launch(const RtgWgradDesc* d, const float* x, const float* dy, float* part, hipStream_t s);

// 0: not served; 1: one input channel; 2: one output channel, long rows; 3: one output channel, short rows;
// 4: two input channels, 3 x 3 (rtg_thin2d.hip)
int rtg_wgrad_thin_kind(const RtgWgradDesc* d) {
  if (rtg_thin2d_wgrad_ok(d)) return 4;
  if (d->groups != 1 || d->C2 != 0 || d->h_k > 1 || d->h_n > 1) return 0;
  if (d->pre_mode != RTG_PRE_NONE && d->pre_mode != RTG_PRE_LRELU) return 0;
  if (d->Cg == 1 && d->C1 == 1 && d->Mg >= 2 && d->Mg <= 32 && (d->K == 5 || d->K == 7 || d->K == 15) &&
      (long long)(kPT - 1) * d->stride + (long long)(d->K - 1) * d->dil + 1 <= 12288)
    return 1;
  if (d->Mg == 1 && d->Cg >= 2 && d->stride == 1 && (d->K == 3 || d->K == 7) && d->pad >= 0 &&
      d->pad <= (d->K - 1) * d->dil && d->dil <= 16 && d->Q <= d->L_in) {
    if (d->Q <= 256 && d->L_in <= 256) return 3;
    if (d->Cg <= 32) return 2;
  }
  return 0;
}

int rtg_wgrad_thin_splits(const RtgWgradDesc* d) {
  const int kind = rtg_wgrad_thin_kind(d);
  if (kind == 0) return RTG_EINVAL;
  if (kind == 4) return rtg_thin2d_wgrad_splits(d);
  if (kind == 3) {
    const int cb = rtg_ceil_div(d->Cg, RTG_THREADS);
    int s = 512 / cb;                                  // ~512 blocks in all
    if (s > d->B) s = d->B;
    return s < 1 ? 1 : s;
  }
  const long long items = (long long)d->B * rtg_ceil_div(kind == 1 ? d->Q : d->L_in, kPT);
  return (int)(items < 512 ? items : 512);
}

int rtg_wgrad_thin_launch(const RtgWgradDesc* d, const float* x, const float* dy, const float* aux, float* part,
                          hipStream_t s) {
  const int kind = rtg_wgrad_thin_kind(d);
  if (kind == 0) return RTG_EINVAL;
  if (kind == 4) return rtg_thin2d_wgrad_launch(d, x, dy, part, s);
  if (d->splits != rtg_wgrad_thin_splits(d)) return RTG_EINVAL;
  if ((d->gy_mode == RTG_PRE_MUL_DLRELU || d->gy_mode == RTG_PRE_MUL_DTANH) && !aux) return RTG_ENULL;
  WtArgs a;
  a.x = x; a.dy = dy; a.aux = aux; a.part = part; a.part_stride = d->part_stride;
  a.B = d->B; a.C = d->Cg; a.L_in = d->L_in; a.M = d->Mg; a.K = d->K; a.stride = d->stride; a.dil = d->dil;
  a.pad = d->pad; a.Q = d->Q; a.dy_L = d->dy_L;
  a.pre_mode = d->pre_mode; a.pre_slope = d->pre_slope;
  a.gy_mode = d->gy_mode; a.gy_slope = d->gy_slope; a.gy_scale = d->gy_scale;
  a.splits = d->splits;
  if (kind == 1) {
    a.n_t = rtg_ceil_div(d->Q, kPT);
    a.items = d->B * a.n_t;
    const size_t lds = ((size_t)(kPT - 1) * d->stride + (size_t)(d->K - 1) * d->dil + 1) * sizeof(float);
#define RTG_WC1(k, rw) RTG_KLAUNCH((wgrad_cin1_kernel<k, rw>), dim3(d->splits), dim3(RTG_THREADS), lds, s, a)
    const int rw = rtg_ceil_div(d->Mg, 4);
    if (rw <= 4) {
      if (d->K == 5) RTG_WC1(5, 4); else if (d->K == 7) RTG_WC1(7, 4); else RTG_WC1(15, 4);
    } else {
      if (d->K == 5) RTG_WC1(5, 8); else if (d->K == 7) RTG_WC1(7, 8); else RTG_WC1(15, 8);
    }
#undef RTG_WC1
    return rtg_launch_status();
  }
  if (kind == 2) {
    a.n_t = rtg_ceil_div(d->L_in, kPT);
    a.items = d->B * a.n_t;
    const size_t lds = ((size_t)kPT + (size_t)(d->K - 1) * d->dil) * sizeof(float);
    if (d->K == 3) RTG_KLAUNCH((wgrad_cout1_long_kernel<3, 8>), dim3(d->splits), dim3(RTG_THREADS), lds, s, a);
    else RTG_KLAUNCH((wgrad_cout1_long_kernel<7, 8>), dim3(d->splits), dim3(RTG_THREADS), lds, s, a);
    return rtg_launch_status();
  }
  const dim3 grid(d->splits, rtg_ceil_div(d->Cg, RTG_THREADS));
  if (d->K == 3) RTG_KLAUNCH((wgrad_cout1_short_kernel<3>), grid, dim3(RTG_THREADS), 0, s, a);
  else RTG_KLAUNCH((wgrad_cout1_short_kernel<7>), grid, dim3(RTG_THREADS), 0, s, a);
  return rtg_launch_status();
}
