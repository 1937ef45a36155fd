// rtg_thin.hip — the two degenerate convolution shapes of the path, as bandwidth kernels instead of MFMA tiles.
//
//   one input channel   conv_pre (generator.py:682), the first conv of every discriminator (discrminator.py:38,157) and
//                       the backward-data of every conv_post: out[b,m,t] = sum_j w[m,j] * x[b,0,t*s - p + j*d].
//                       K multiply-adds per output: the kernel is bound by writing `out` once.
//   one output channel  conv_post of G and of every discriminator (generator.py:722, discrminator.py:45,163) and the
//                       backward-data of the C_in = 1 convs: out[b,0,t] = sum_c sum_j w[c,j] * pre(x[b,c,t*s - p + j*d]).
//                       Bound by reading `x` once.
// On the matrix cores these shapes fill 1/16 of a tile and (worse) produce a handful of workgroups that each walk all
// channel chunks serially: 120 us for 27 MB of input.  Here every output position is a lane (coalesced along the
// row), the waves of a block split the rows / channels, and nothing is staged but the (tiny) weight tensor.
// Both kernels read the weights from the SAME packed buffers the MFMA kernels use (rtg_weights_pack), so callers see
// no difference: rtg_conv1d dispatches here (rtg_conv1d.hip).
#include "rtg_common.h"

namespace {

struct ThinArgs {
  const float *x, *aux, *wp, *bias, *mask, *res;
  float* out;
  int B, C, L_in, M, K, stride, dil, pad, Q, out_L;
  int pre_mode;
  float pre_slope, mask_slope, out_scale;
  int act;
  float act_slope;
  int tile_m, tap_major;
  long long n_pos;                 // B * Q
};

// index of logical weight (row m, channel c of Cg, tap) in the packed layouts of rtg_weights_pack (groups == 1)
__device__ __forceinline__ long long packed_index(int m, int c, int tap, int Cg, int K, int TM, int tap_major) {
  const int KK = 64 / TM, CPN = RTG_CK / KK;
  const int mt = m / TM, mr = m - mt * TM;
  if (tap_major) {
    const int TG = (K + KK - 1) / KK;
    const int n_grp = (Cg * TG + CPN - 1) / CPN;
    const int ks = c * TG + tap / KK, kk = tap % KK;
    const int grp = ks / CPN, cp = ks - grp * CPN;
    return (((long long)mt * n_grp + grp) * CPN + cp) * (KK * TM) + kk * TM + mr;
  }
  const int n_cc = (Cg + RTG_CK - 1) / RTG_CK;
  const int cc = c / RTG_CK, cl = c - cc * RTG_CK;
  const int cp = cl / KK, kk = cl - cp * KK;
  return ((((long long)mt * n_cc + cc) * K + tap) * CPN + cp) * (KK * TM) + kk * TM + mr;
}

__device__ __forceinline__ float thin_act(float v, int act, float slope) {
  if (act == RTG_ACT_LRELU) return fmaf(fminf(v, 0.f), slope, fmaxf(v, 0.f));
  if (act == RTG_ACT_TANH) return tanhf(v);
  return v;
}

constexpr int kMaxTaps = 16;       // one-input-channel kernel keeps the row's taps in registers
constexpr int kRowsPerBlock = 32;  // output rows one block of the one-input-channel kernel produces

// ---- one input channel: lane = output position (flattened over clips), block = 256 positions x 32 rows
__global__ __launch_bounds__(RTG_THREADS) void thin_cin1_kernel(const ThinArgs a) {
  __shared__ float w[kRowsPerBlock * kMaxTaps];
  __shared__ float bs[kRowsPerBlock];
  const int m0 = blockIdx.y * kRowsPerBlock;
  const int rows = min(kRowsPerBlock, a.M - m0);
  for (int e = threadIdx.x; e < rows * a.K; e += RTG_THREADS) {
    const int r = e / a.K, j = e - r * a.K;
    w[r * kMaxTaps + j] = a.wp[packed_index(m0 + r, 0, j, 1, a.K, a.tile_m, a.tap_major)];
  }
  for (int r = threadIdx.x; r < rows; r += RTG_THREADS) bs[r] = a.bias ? a.bias[m0 + r] : 0.f;
  __syncthreads();
  const long long n = (long long)blockIdx.x * RTG_THREADS + threadIdx.x;
  if (n >= a.n_pos) return;
  const int b = (int)(n / a.Q), t = (int)(n - (long long)b * a.Q);
  const float* xr = a.x + (size_t)b * a.L_in;
  const float* ar = a.aux ? a.aux + (size_t)b * a.L_in : nullptr;
  float xv[kMaxTaps];
#pragma unroll
  for (int j = 0; j < kMaxTaps; ++j) {
    const int pos = t * a.stride - a.pad + j * a.dil;
    float v = 0.f;
    if (j < a.K && pos >= 0 && pos < a.L_in) {
      v = xr[pos];
      if (a.pre_mode == RTG_PRE_LRELU) v = fmaf(fminf(v, 0.f), a.pre_slope, fmaxf(v, 0.f));
      else if (a.pre_mode == RTG_PRE_MUL_DLRELU) v *= (ar[pos] > 0.f ? 1.f : a.pre_slope);
      else if (a.pre_mode == RTG_PRE_MUL_DTANH) v *= fmaf(-ar[pos], ar[pos], 1.f);
    }
    xv[j] = v;
  }
  const float mslope = a.mask ? a.mask_slope : 1.f;
  for (int r = 0; r < rows; ++r) {
    float acc = bs[r];
#pragma unroll
    for (int j = 0; j < kMaxTaps; ++j)
      if (j < a.K) acc = fmaf(w[r * kMaxTaps + j], xv[j], acc);
    const size_t o = ((size_t)b * a.M + m0 + r) * a.out_L + t;
    if (a.mask) acc *= (a.mask[o] > 0.f ? 1.f : mslope);
    if (a.res) acc += a.res[o];
    a.out[o] = thin_act(acc * a.out_scale, a.act, a.act_slope);
  }
}

// ---- one output channel: lane = output position (flattened over clips), the 8 waves split the input channels.
// Loads go through a buffer descriptor (out-of-range taps get an offset past the records and read 0), four channels
// x K taps in flight per wave: the kernel lives on memory-level parallelism, there is no reuse to stage.
constexpr int kCoutWaves = 8;
using rsrc_t = __amdgpu_buffer_rsrc_t;

__global__ __launch_bounds__(64 * kCoutWaves) void thin_cout1_kernel(const ThinArgs a) {
  extern __shared__ float sm[];                 // [C*K] weights in logical order, then [waves][64] partial sums
  float* w = sm;
  float* part = sm + a.C * a.K;
  for (int e = threadIdx.x; e < a.C * a.K; e += 64 * kCoutWaves) {
    const int c = e / a.K, j = e - c * a.K;
    w[e] = a.wp[packed_index(0, c, j, a.C, a.K, a.tile_m, a.tap_major)];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long long n = (long long)blockIdx.x * 64 + lane;
  const bool live = n < a.n_pos;
  const int b = live ? (int)(n / a.Q) : 0, t = live ? (int)(n - (long long)b * a.Q) : 0;
  const int p0 = t * a.stride - a.pad;
  const int x_bytes = a.B * a.C * a.L_in * 4;
  const rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, x_bytes, 0x00020000);
  const rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)(a.aux ? a.aux : a.x), 0, a.aux ? x_bytes : 0, 0x00020000);
  const unsigned clip_off = (unsigned)b * (unsigned)a.C * (unsigned)a.L_in * 4u;
  const unsigned row_bytes = (unsigned)a.L_in * 4u;
  float acc = 0.f;
  auto one = [&](int c, int j) __attribute__((always_inline)) {
    const int pos = p0 + j * a.dil;
    const bool ok = live && c < a.C && pos >= 0 && pos < a.L_in;
    const unsigned off = ok ? clip_off + (unsigned)c * row_bytes + (unsigned)pos * 4u : 0x80000000u;
    float v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, off, 0, 0));
    if (a.pre_mode == RTG_PRE_LRELU) {
      v = fmaf(fminf(v, 0.f), a.pre_slope, fmaxf(v, 0.f));
    } else if (a.pre_mode >= RTG_PRE_MUL_DLRELU) {
      const float av = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ra, off, 0, 0));
      v *= (a.pre_mode == RTG_PRE_MUL_DTANH) ? fmaf(-av, av, 1.f) : fmaf(1.f - a.pre_slope, (float)(av > 0.f), a.pre_slope);
    }
    return v;
  };
  for (int c0 = wave * 4; c0 < a.C; c0 += 4 * kCoutWaves) {
    for (int j = 0; j < a.K; ++j) {
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = one(c0 + u, j);
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (c0 + u < a.C) acc = fmaf(w[(c0 + u) * a.K + j], v[u], acc);
    }
  }
  part[wave * 64 + lane] = acc;
  __syncthreads();
  if (wave == 0 && live) {
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < kCoutWaves; ++k) v += part[k * 64 + lane];                       // fixed order
    if (a.bias) v += a.bias[0];
    const size_t o = (size_t)b * a.out_L + t;
    if (a.mask) v *= (a.mask[o] > 0.f ? 1.f : a.mask_slope);
    if (a.res) v += a.res[o];
    a.out[o] = thin_act(v * a.out_scale, a.act, a.act_slope);
  }
}

}  // namespace

// 0: not a thin shape (use the MFMA kernel), 1: one input channel, 2: one output channel
int rtg_thin_kind(const RtgConv1dDesc* d) {
  if (d->groups != 1 || d->C2 != 0 || d->shuf_S != 1 || d->out_split != 0 || d->accumulate) return 0;
  if (d->h_k > 1 || d->h_n > 1) return 0;
  if (d->Cg == 1 && d->K <= kMaxTaps) return 1;
  if (d->Mg == 1 && (long long)d->Cg * d->K <= 12288 && (long long)d->B * d->Cg * d->L_in * 4 < (1ll << 31)) return 2;
  return 0;
}

int rtg_thin_launch(int kind, const RtgConv1dDesc* d, const float* x, const float* aux, const float* wp,
                    const float* bias, const float* mask, const float* res, float* out, hipStream_t s) {
  ThinArgs a;
  a.x = x; a.aux = (d->pre_mode >= RTG_PRE_MUL_DLRELU) ? aux : nullptr; a.wp = wp; a.bias = bias; a.mask = mask;
  a.res = res; a.out = out;
  a.B = d->B; a.C = d->Cg; a.L_in = d->L_in; a.M = d->Mg; a.K = d->K; a.stride = d->stride; a.dil = d->dil;
  a.pad = d->pad; a.Q = d->Q; a.out_L = d->out_L; a.pre_mode = d->pre_mode; a.pre_slope = d->pre_slope;
  a.mask_slope = d->mask_slope; a.out_scale = d->out_scale; a.act = d->act; a.act_slope = d->act_slope;
  a.tile_m = d->tile_m; a.tap_major = d->tap_major ? 1 : 0;
  a.n_pos = (long long)d->B * d->Q;
  if (kind == 1) {
    const long long gx = (a.n_pos + RTG_THREADS - 1) / RTG_THREADS;
    const int gy = rtg_ceil_div(d->Mg, kRowsPerBlock);
    if (gx > 0x7fffffffLL || gy > 65535) return RTG_ERANGE;
    RTG_KLAUNCH(thin_cin1_kernel, dim3((unsigned)gx, gy), dim3(RTG_THREADS), 0, s, a);
  } else {
    const long long gx = (a.n_pos + 63) / 64;
    if (gx > 0x7fffffffLL) return RTG_ERANGE;
    const size_t lds = ((size_t)d->Cg * d->K + 64 * kCoutWaves) * sizeof(float);
    RTG_KLAUNCH(thin_cout1_kernel, dim3((unsigned)gx), dim3(64 * kCoutWaves), lds, s, a);
  }
  return rtg_launch_status();
}
