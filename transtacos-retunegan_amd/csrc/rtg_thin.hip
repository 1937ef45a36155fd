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
#include <stdlib.h>

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
  if (act == RTG_ACT_LRELU) return v > 0.f ? v : v * slope;     // (a NaN stays a NaN: the step's NaN guard relies on it)
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
      if (a.pre_mode == RTG_PRE_LRELU) v = v > 0.f ? v : v * a.pre_slope;
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
  extern __shared__ __attribute__((aligned(16))) float sm[];                 // [C*K] weights in logical order, then [waves][64] partial sums
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
      v = v > 0.f ? v : v * a.pre_slope;
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


// =====================================================================================================================
// Tile-staged bandwidth kernels (round 2).  The two kernels above fetch every tap with its own dword load and work on
// 64..256 positions per block; measured 405 GB/s (one output channel) and 1 047 GB/s (one input channel).  The
// kernels below move every input byte ONCE with wide coalesced loads into LDS (activation applied on the way), take
// the K taps from LDS, and write 16 bytes per lane.  They serve the shapes the train step actually has; anything else
// (aux operand on the one-output-channel shape, rows beyond the staging limits) falls back to the kernels above.
// =====================================================================================================================
struct TileGeo {
  int PT, TP, tp_shift, R, NG, CH, W, Wp, tiles, w_floats, vec;
  int TPRow, tprow_shift, unit, shift;
  int RB;                      // one-input-channel kernels: output rows per block
  int chunks;                  // flat kernel: chunks of kFlatChunk outputs per clip
};

__device__ __forceinline__ float thin_pre(float v, const ThinArgs& a) {
  return a.pre_mode == RTG_PRE_LRELU ? (v > 0.f ? v : v * a.pre_slope) : v;
}

// ---- one output channel, long rows.  Block = (clip, tile of PT positions); the C input rows are staged CH at a time as
// [CH][window of the tile] — float4 loads from 16-byte aligned row addresses (VEC) or dword loads, no integer division
// (thread = column q of row rr, rows rr, rr + RP, ...), up to 8 loads in flight per thread — with the activation applied
// on the way and the window start at LDS column 0 of each row.  Compute: thread = (4 consecutive positions, channel
// group); for stride 1 / dilation 1 the 4 + K - 1 inputs of a channel come from aligned 16-byte LDS reads and live in
// registers (K weights + 4K multiply-adds per 3..5 LDS reads); other strides take scalar LDS reads.  The NG partial sums
// of a position meet in LDS and are added in fixed order.
template <bool VEC, int KT>
__global__ __launch_bounds__(RTG_THREADS) void cout1_long_kernel(const ThinArgs a, const TileGeo g) {
  const int K = KT > 0 ? KT : a.K;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* wl = sm;                                     // [C][Kp]
  float* xs = sm + g.w_floats;                        // [CH][Wp]
  const int tid = threadIdx.x;
  const int b = blockIdx.x / g.tiles, tile = blockIdx.x - b * g.tiles;
  // g.shift: the tile grid starts `shift` outputs left of output 0 so that (stride 1) every window starts on a 16-byte
  // boundary of the input row: m == 0, aligned 16-byte LDS writes and reads
  const int t0 = tile * g.PT - g.shift;
  const int g0 = t0 * a.stride - a.pad;
  const int a0 = VEC ? (g0 & ~3) : g0;               // floor to a multiple of 4 (also for negative g0)
  const int m = g0 - a0;
  const int Kp = (K + 3) & ~3;
  for (int e = tid; e < a.C * Kp; e += RTG_THREADS) {
    const int c = e / Kp, j = e - c * Kp;
    wl[e] = j < K ? a.wp[packed_index(0, c, j, a.C, K, a.tile_m, a.tap_major)] : 0.f;
  }
  const float* xb = a.x + (size_t)b * a.C * a.L_in;
  const int pl = tid & (g.TP - 1), cg = tid >> g.tp_shift;
  const int NQ = VEC ? ((m + g.W + 3) >> 2) : g.W;   // load columns per row (float4 / dword)
  // staging: thread = column qcol (+ TPRow * pass) of rows rr, rr + RP, ...; everything that does not depend on the row
  // is computed once, the row loop advances two pointers
  const int qcol = tid & (g.TPRow - 1), rr = tid >> g.tprow_shift;
  const int RP = RTG_THREADS >> g.tprow_shift;
  const int ncp = (NQ + g.TPRow - 1) >> g.tprow_shift;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int c0 = 0; c0 < a.C; c0 += g.CH) {
    const int chc = min(g.CH, a.C - c0);
    __syncthreads();
    for (int cp = 0; cp < ncp; ++cp) {
      const int q = qcol + (cp << g.tprow_shift);
      const int pos = VEC ? a0 + 4 * q : g0 + q;
      const bool col_ok = q < NQ;
      const bool in_row = col_ok && pos >= 0 && pos < a.L_in;
      const int col = VEC ? 4 * q - m : q;           // LDS column of the first loaded float (window start = column 0)
      const float* src = xb + (size_t)(c0 + rr) * a.L_in + pos;
      float* dst = xs + rr * g.Wp + col;
      const size_t src_step = (size_t)RP * a.L_in;
      const int dst_step = RP * g.Wp;
      for (int c = rr; c < chc; c += 8 * RP) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (in_row && c + u * RP < chc) {
            if (VEC) v[u] = *reinterpret_cast<const f32x4*>(src + u * src_step);
            else v[u].x = src[u * src_step];
          }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if (!col_ok || c + u * RP >= chc) continue;
          float* d_ = dst + u * dst_step;
          if (VEC && m == 0) {
            f32x4 t = v[u];
            t.x = thin_pre(t.x, a); t.y = thin_pre(t.y, a); t.z = thin_pre(t.z, a); t.w = thin_pre(t.w, a);
            *reinterpret_cast<f32x4*>(d_) = t;
          } else if (VEC) {
            const float t4[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
            for (int e = 0; e < 4; ++e)               // the floats left of the window (first float4 of a row) are dropped
              if (col + e >= 0) d_[e] = thin_pre(t4[e], a);
          } else {
            d_[0] = thin_pre(v[u].x, a);
          }
        }
        src += 8 * src_step;
        dst += 8 * dst_step;
      }
    }
    __syncthreads();
    if (g.unit) {
      for (int c = cg; c < chc; c += g.NG) {
        const f32x4* row4 = reinterpret_cast<const f32x4*>(xs + c * g.Wp) + pl;
        const f32x4* w4 = reinterpret_cast<const f32x4*>(wl + (c0 + c) * Kp);
        float xw[20], w[16];
#pragma unroll
        for (int q = 0; q < 5; ++q) {
          if (4 * q < 3 + K) {
            const f32x4 t = row4[q];
            xw[4 * q] = t.x; xw[4 * q + 1] = t.y; xw[4 * q + 2] = t.z; xw[4 * q + 3] = t.w;
          } else {
            xw[4 * q] = xw[4 * q + 1] = xw[4 * q + 2] = xw[4 * q + 3] = 0.f;
          }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (4 * q < K) {
            const f32x4 t = w4[q];
            w[4 * q] = t.x; w[4 * q + 1] = t.y; w[4 * q + 2] = t.z; w[4 * q + 3] = t.w;
          } else {
            w[4 * q] = w[4 * q + 1] = w[4 * q + 2] = w[4 * q + 3] = 0.f;
          }
        }
#pragma unroll
        for (int j = 0; j < kMaxTaps; ++j) {
          if (j < K) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = fmaf(w[j], xw[i + j], acc[i]);
          }
        }
      }
    } else {
      for (int c = cg; c < chc; c += g.NG) {
        const float* row = xs + c * g.Wp + 4 * pl * a.stride;
        const float* wr = wl + (c0 + c) * Kp;
        for (int j = 0; j < K; ++j) {
          const float w = wr[j];
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[i] = fmaf(w, row[i * a.stride + j * a.dil], acc[i]);
        }
      }
    }
  }
  __syncthreads();
  float* part = xs;                                   // [NG][PT]
#pragma unroll
  for (int i = 0; i < 4; ++i) part[cg * g.PT + 4 * pl + i] = acc[i];
  __syncthreads();
  for (int e = tid; e < g.PT; e += RTG_THREADS) {
    const int t = t0 + e;
    if (t < 0 || t >= a.Q) continue;
    float v = part[e];
    for (int k = 1; k < g.NG; ++k) v += part[k * g.PT + e];                          // fixed order
    if (a.bias) v += a.bias[0];
    const size_t o = (size_t)b * a.out_L + t;
    if (a.mask) v *= (a.mask[o] > 0.f ? 1.f : a.mask_slope);
    if (a.res) v += a.res[o];
    a.out[o] = thin_act(v * a.out_scale, a.act, a.act_slope);
  }
}

// ---- one output channel, short rows (the whole row of <= 256 outputs is one tile): the [CH, L_in] input rows of a clip
// are one contiguous run in memory and are copied flat (16-byte loads when the run is aligned), zero padding is a
// predicate in the tap loop.  512 threads = (position, channel group); the loads of chunk i+1 are in flight while chunk i
// is computed (one block per clip: there are only B blocks, so the overlap has to happen inside the block).
constexpr int kShortThreads = 512;
constexpr int kShortRegs = 6;                         // float4 (or dword x 4) registers of the prefetch per thread
template <bool VEC>
__global__ __launch_bounds__(kShortThreads) void cout1_short_kernel(const ThinArgs a, const TileGeo g) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* wl = sm;                                     // [C][K]
  float* xs = sm + g.w_floats;                        // [CH][L_in] flat
  const int tid = threadIdx.x;
  const int b = blockIdx.x;
  for (int e = tid; e < a.C * a.K; e += kShortThreads) {
    const int c = e / a.K, j = e - c * a.K;
    wl[e] = a.wp[packed_index(0, c, j, a.C, a.K, a.tile_m, a.tap_major)];
  }
  const float* xb = a.x + (size_t)b * a.C * a.L_in;
  const int pl = tid & (g.TP - 1), cg = tid >> g.tp_shift;
  const int p0 = pl * a.stride - a.pad;
  float acc = 0.f;
  f32x4 v[kShortRegs];
  auto issue = [&](int c0) __attribute__((always_inline)) {
    const int n = min(g.CH, a.C - c0) * a.L_in;
    const float* src = xb + (size_t)c0 * a.L_in;
#pragma unroll
    for (int u = 0; u < kShortRegs; ++u) {
      const int f = tid + u * kShortThreads;
      v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (VEC) {
        if (4 * f < n) v[u] = reinterpret_cast<const f32x4*>(src)[f];
      } else {
        if (4 * f < n) v[u].x = src[4 * f];
        if (4 * f + 1 < n) v[u].y = src[4 * f + 1];
        if (4 * f + 2 < n) v[u].z = src[4 * f + 2];
        if (4 * f + 3 < n) v[u].w = src[4 * f + 3];
      }
    }
  };
  issue(0);
  for (int c0 = 0; c0 < a.C; c0 += g.CH) {
    const int chc = min(g.CH, a.C - c0);
    __syncthreads();                                  // (first pass: the weights; later: the previous chunk is consumed)
#pragma unroll
    for (int u = 0; u < kShortRegs; ++u) {
      const int f = tid + u * kShortThreads;
      if (4 * f < chc * a.L_in) {                     // (a partial last float4 is zero-filled; xs has the slack)
        f32x4 t = v[u];
        t.x = thin_pre(t.x, a); t.y = thin_pre(t.y, a); t.z = thin_pre(t.z, a); t.w = thin_pre(t.w, a);
        reinterpret_cast<f32x4*>(xs)[f] = t;
      }
    }
    __syncthreads();
    if (c0 + g.CH < a.C) issue(c0 + g.CH);
    if (pl < a.Q) {
      for (int c = cg; c < chc; c += g.NG) {
        const float* row = xs + c * a.L_in;
        const float* wr = wl + (c0 + c) * a.K;
        for (int j = 0; j < a.K; ++j) {
          const int pos = p0 + j * a.dil;
          if (pos >= 0 && pos < a.L_in) acc = fmaf(wr[j], row[pos], acc);
        }
      }
    }
  }
  __syncthreads();
  float* part = xs;                                   // [NG][TP]
  part[cg * g.TP + pl] = acc;
  __syncthreads();
  if (tid < a.Q) {
    float s_ = part[tid];
    for (int k = 1; k < g.NG; ++k) s_ += part[k * g.TP + tid];                      // fixed order
    if (a.bias) s_ += a.bias[0];
    const size_t o = (size_t)b * a.out_L + tid;
    if (a.mask) s_ *= (a.mask[o] > 0.f ? 1.f : a.mask_slope);
    if (a.res) s_ += a.res[o];
    a.out[o] = thin_act(s_ * a.out_scale, a.act, a.act_slope);
  }
}

// ---- one output channel, k3 "same" (conv_post of every discriminator: 512 -> 1, rows of 10..128 positions).  The clips
// are 20-260 KB each and there are 64-704 of them: a block per clip (cout1_short) leaves the chip latency bound (27-47 us
// for 14-40 MB).  Here the 64 lanes of a wave are 64 (clip, position) columns — several whole clips side by side when
// the rows are short, a 64-position block of a clip when they are long — and the 8 waves of a block split the input
// channels: a wave requests its C / 8 channels' three samples per column (16 channels = 48 loads in flight, the
// neighbours come from the same cache lines), the 8 partial sums of a column meet in LDS and are added in fixed order.
constexpr int kK3Waves = 8;
__global__ __launch_bounds__(64 * kK3Waves) void cout1_k3_kernel(const ThinArgs a, int cpb, int bpc) {
  // cpb: clips per block (rows of <= 64 positions), bpc: position blocks per clip (longer rows; then cpb == 1)
  __shared__ float wl[3 * 1024];                      // [C][3] (C <= 1024)
  __shared__ float part[kK3Waves][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int e = tid; e < a.C * 3; e += 64 * kK3Waves) {
    const int c = e / 3, j = e - c * 3;
    wl[e] = a.wp[packed_index(0, c, j, a.C, 3, a.tile_m, a.tap_major)];
  }
  int clip, pos;
  bool col_ok;
  if (bpc == 1) {
    const int sub = lane / a.L_in;                    // (L_in == Q: "same" conv)
    clip = blockIdx.x * cpb + sub;
    pos = lane - sub * a.L_in;
    col_ok = sub < cpb && clip < a.B;
  } else {
    clip = blockIdx.x / bpc;
    pos = (blockIdx.x - clip * bpc) * 64 + lane;
    col_ok = pos < a.L_in;
  }
  const rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.B * a.C * a.L_in * 4, 0x00020000);
  const int cps = a.C / kK3Waves;                     // channels per wave
  const int c0 = wave * cps;
  unsigned off[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int p = pos + j - 1;
    off[j] = (col_ok && p >= 0 && p < a.L_in) ? (unsigned)((clip * a.C + c0) * a.L_in + p) * 4u : 0x80000000u;
  }
  __syncthreads();
  float acc = 0.f;
  const unsigned rowb = (unsigned)a.L_in * 4u;
  for (int cb = 0; cb < cps; cb += 16) {
    float v[16][3];
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j)
        v[i][j] = (cb + i < cps) ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, off[j] + (unsigned)(cb + i) * rowb, 0, 0))
                                 : 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (cb + i < cps) {
        const float* w = wl + (c0 + cb + i) * 3;
#pragma unroll
        for (int j = 0; j < 3; ++j) acc = fmaf(w[j], thin_pre(v[i][j], a), acc);
      }
    }
  }
  part[wave][lane] = acc;
  __syncthreads();
  if (wave == 0 && col_ok) {
    float s_ = part[0][lane];
#pragma unroll
    for (int k = 1; k < kK3Waves; ++k) s_ += part[k][lane];                          // fixed order
    if (a.bias) s_ += a.bias[0];
    const size_t o = (size_t)clip * a.out_L + pos;
    if (a.mask) s_ *= (a.mask[o] > 0.f ? 1.f : a.mask_slope);
    if (a.res) s_ += a.res[o];
    a.out[o] = thin_act(s_ * a.out_scale, a.act, a.act_slope);
  }
}

// ---- the same operator for rows of <= 64 positions (round 4): conv_post of the period discriminators (rows of 10 .. 34
// positions, 192-704 clips) and of the lower MSD scales.  cout1_k3_kernel makes a lane a (clip, position) column: with rows
// of 10 floats a wave-wide load touches 6 clips' 40-byte segments (256 useful bytes from six cache lines per instruction,
// three instructions per channel for the taps): 0.5-0.9 TB/s, bound by issuing ~250 vector-memory instructions per clip.
// A clip's [C][L] block is CONTIGUOUS, so here a wave reads it as it lies: lane = (row r < R, position q) of R = 64 / L
// consecutive channel rows — one coalesced load instruction per R rows, every element read ONCE — and keeps the three
// per-tap channel sums T_j[q] = sum_c w[c][j] * x[c][q] of its column (out[q] = T_0[q-1] + T_1[q] + T_2[q+1]); the 8 waves
// of a block split the channels, the partial sums meet in LDS and are added in fixed order (wave, then row).
__global__ __launch_bounds__(64 * kK3Waves) void cout1_k3_rows_kernel(const ThinArgs a) {
  __shared__ float wl[3 * 1024];                      // [C][3] (C <= 1024)
  __shared__ float part[kK3Waves][64][3];
  __shared__ float tsum[3][64 + 2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int L = a.L_in, C = a.C;
  for (int e = tid; e < C * 3; e += 64 * kK3Waves) {
    const int c = e / 3, j = e - c * 3;
    wl[e] = a.wp[packed_index(0, c, j, C, 3, a.tile_m, a.tap_major)];
  }
  const int R = 64 / L;                               // channel rows per wave-wide load
  const int r = lane / L, q = lane - r * L;
  const bool lane_ok = r < R;
  const int clip = blockIdx.x;
  const int cps = C / kK3Waves, c0 = wave * cps;      // this wave's channels
  const rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.B * C * L * 4, 0x00020000);
  const unsigned base = (unsigned)((clip * C + c0) * L) * 4u;
  __syncthreads();
  float t0 = 0.f, t1 = 0.f, t2 = 0.f;
  constexpr int U = 8;                                // loads in flight per lane
  for (int k0 = 0; k0 < cps; k0 += U * R) {
    float v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int cl = k0 + u * R + r;                  // channel within the wave's range
      const unsigned off = (lane_ok && cl < cps) ? base + (unsigned)(cl * L + q) * 4u : 0x80000000u;
      v[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, off, 0, 0));
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int cl = k0 + u * R + r;
      if (lane_ok && cl < cps) {
        const float* w = wl + (c0 + cl) * 3;
        const float xv = thin_pre(v[u], a);
        t0 = fmaf(w[0], xv, t0);
        t1 = fmaf(w[1], xv, t1);
        t2 = fmaf(w[2], xv, t2);
      }
    }
  }
  part[wave][lane][0] = t0; part[wave][lane][1] = t1; part[wave][lane][2] = t2;
  __syncthreads();
  if (tid < 3 * (L + 2)) {                             // T_j[-1 .. L]: zero outside the row
    const int j = tid / (L + 2), p = tid - j * (L + 2) - 1;
    float s_ = 0.f;
    if (p >= 0 && p < L)
      for (int w = 0; w < kK3Waves; ++w)
        for (int rr = 0; rr < R; ++rr) s_ += part[w][rr * L + p][j];                    // fixed order
    tsum[j][p + 1] = s_;
  }
  __syncthreads();
  if (tid < L) {
    float s_ = tsum[0][tid] + tsum[1][tid + 1] + tsum[2][tid + 2];                      // T_0[q-1] + T_1[q] + T_2[q+1]
    if (a.bias) s_ += a.bias[0];
    const size_t o = (size_t)clip * a.out_L + tid;
    if (a.mask) s_ *= (a.mask[o] > 0.f ? 1.f : a.mask_slope);
    if (a.res) s_ += a.res[o];
    a.out[o] = thin_act(s_ * a.out_scale, a.act, a.act_slope);
  }
}

// ---- one output channel, 3 x 3 "same" in two dimensions (conv_post of the STFT discriminators, discrminator.py:262:
// Conv2d(512, 1, (3, 3), padding (1, 1)) on maps of 29 x 5 .. 8 x 18).  On the matrix-core path one of 16 tile rows is
// used and a launch takes 220 us (0.2-0.4 TFLOP/s, 2.6 ms of a full-stack step).  Same scheme as cout1_k3_kernel: a lane
// is one (item, row, column) output, the 8 waves of a block split the input channels (8 channels = 72 loads in flight
// per lane, neighbours from the same cache lines), fixed-order LDS reduction.
__global__ __launch_bounds__(64 * kK3Waves) void cout1_k3x3_kernel(const ThinArgs a, int C, int H, int items) {
  __shared__ float wl[9 * 512];                       // [C][kh][kw] (C <= 512)
  __shared__ float part[kK3Waves][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int W = a.L_in;
  for (int e = tid; e < C * 9; e += 64 * kK3Waves) {
    const int c = e / 9, r = e - c * 9;
    // the 1-D operator's channels are (c, kernel row) pairs, its taps the kernel columns
    wl[e] = a.wp[packed_index(0, c * 3 + r / 3, r % 3, C * 3, 3, a.tile_m, a.tap_major)];
  }
  const int col = blockIdx.x * 64 + lane;
  const int hw = H * W;
  const int item = col / hw;
  const int rem = col - item * hw;
  const int h = rem / W, w = rem - h * W;
  const bool col_ok = item < items;
  const rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, items * C * hw * 4, 0x00020000);
  const int cps = C / kK3Waves;
  const int c0 = wave * cps;
  unsigned off[9];
#pragma unroll
  for (int r = 0; r < 9; ++r) {
    const int hh = h + r / 3 - 1, ww = w + r % 3 - 1;
    off[r] = (col_ok && hh >= 0 && hh < H && ww >= 0 && ww < W) ? (unsigned)(((item * C + c0) * H + hh) * W + ww) * 4u
                                                                 : 0x80000000u;
  }
  __syncthreads();
  float acc = 0.f;
  const unsigned planeb = (unsigned)hw * 4u;
  for (int cb = 0; cb < cps; cb += 8) {
    float v[8][9];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int r = 0; r < 9; ++r)
        v[i][r] = (cb + i < cps) ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, off[r] + (unsigned)(cb + i) * planeb, 0, 0))
                                 : 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (cb + i < cps) {
        const float* wq = wl + (c0 + cb + i) * 9;
#pragma unroll
        for (int r = 0; r < 9; ++r) acc = fmaf(wq[r], thin_pre(v[i][r], a), acc);
      }
    }
  }
  part[wave][lane] = acc;
  __syncthreads();
  if (wave == 0 && col_ok) {
    float s_ = part[0][lane];
#pragma unroll
    for (int k = 1; k < kK3Waves; ++k) s_ += part[k][lane];                          // fixed order
    if (a.bias) s_ += a.bias[0];
    if (a.mask) s_ *= (a.mask[col] > 0.f ? 1.f : a.mask_slope);
    if (a.res) s_ += a.res[col];
    a.out[col] = thin_act(s_ * a.out_scale, a.act, a.act_slope);
  }
}

// stage the input window [g0, g0 + W) of clip b (one channel) into xs, pre-activation and aux factors applied
__device__ __forceinline__ void cin1_stage(const ThinArgs& a, int b, int g0, int W, float* xs) {
  const float* xr = a.x + (size_t)b * a.L_in;
  const float* ar = a.aux ? a.aux + (size_t)b * a.L_in : nullptr;
  for (int i0 = threadIdx.x; i0 < W; i0 += 4 * RTG_THREADS) {
    float v[4], av[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int pos = g0 + i0 + u * RTG_THREADS;
      const bool ok = i0 + u * RTG_THREADS < W && pos >= 0 && pos < a.L_in;
      v[u] = ok ? xr[pos] : 0.f;
      av[u] = (ok && ar) ? ar[pos] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * RTG_THREADS;
      if (i >= W) continue;
      float t = v[u];
      if (a.pre_mode == RTG_PRE_LRELU) t = t > 0.f ? t : t * a.pre_slope;
      else if (a.pre_mode == RTG_PRE_MUL_DLRELU) t *= (av[u] > 0.f ? 1.f : a.pre_slope);
      else if (a.pre_mode == RTG_PRE_MUL_DTANH) t *= fmaf(-av[u], av[u], 1.f);
      xs[i] = t;
    }
  }
}

// ---- one input channel, long rows: block = (clip, 1024 positions) x RB output rows; a thread keeps the 4 x K input values
// of its four positions in registers and walks the rows: K broadcast weight reads, 4K multiply-adds per row.
// VEC (out_L % 4 == 0, 16-byte aligned tensors): the four positions are consecutive, ONE 16-byte store per row.
// !VEC (rows of any length and alignment — the first layer of the period discriminators: rows of 2731 / 1639 / 1171 / 745,
// discrminator.py:100): the four positions are 256 apart, four 4-byte stores per row, each coalesced over the wave.  These
// rows used to take cin1_flat_kernel, whose blocks stage the clip's WHOLE input row (8193 samples) for 8192 outputs.
constexpr int kRowTile = 4 * RTG_THREADS;
template <int KT, bool VEC>
__global__ __launch_bounds__(RTG_THREADS) void cin1_rows_kernel(const ThinArgs a, const TileGeo g) {
  const int K = KT > 0 ? KT : a.K;
  constexpr int KR = KT > 0 ? KT : kMaxTaps;         // taps kept in registers
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* wl = sm;                                     // [RB][kMaxTaps]
  float* bl = wl + g.RB * kMaxTaps;                   // [RB]
  float* xs = bl + g.RB;                              // [W]
  const int tid = threadIdx.x;
  const int b = blockIdx.x / g.tiles, tile = blockIdx.x - b * g.tiles;
  const int t0 = tile * kRowTile;
  const int m0 = blockIdx.y * g.RB;
  const int rows = min(g.RB, a.M - m0);
  for (int e = tid; e < rows * kMaxTaps; e += RTG_THREADS) {
    const int r = e / kMaxTaps, j = e - r * kMaxTaps;
    wl[e] = j < K ? a.wp[packed_index(m0 + r, 0, j, 1, K, a.tile_m, a.tap_major)] : 0.f;
  }
  for (int r = tid; r < rows; r += RTG_THREADS) bl[r] = a.bias ? a.bias[m0 + r] : 0.f;
  cin1_stage(a, b, t0 * a.stride - a.pad, g.W, xs);
  __syncthreads();
  // position i of this thread within the tile: 4 tid + i (VEC) or tid + 256 i
  constexpr int PS = VEC ? 1 : RTG_THREADS, PB = VEC ? 4 : 1;
  const int t = t0 + PB * tid;
  if (t >= a.Q) return;
  float xv[4][KR];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < KR; ++j) xv[i][j] = j < K ? xs[(PB * tid + PS * i) * a.stride + j * a.dil] : 0.f;
  const float mslope = a.mask_slope;
  for (int r = 0; r < rows; ++r) {
    const f32x4* w4 = reinterpret_cast<const f32x4*>(wl + r * kMaxTaps);
    float w[kMaxTaps];
#pragma unroll
    for (int q = 0; q < (KR + 3) / 4; ++q) {
      const f32x4 t4 = w4[q];
      w[4 * q] = t4.x; w[4 * q + 1] = t4.y; w[4 * q + 2] = t4.z; w[4 * q + 3] = t4.w;
    }
    float acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float s_ = bl[r];
#pragma unroll
      for (int j = 0; j < KR; ++j)
        if (j < K) s_ = fmaf(w[j], xv[i][j], s_);
      acc[i] = s_;
    }
    const size_t o = ((size_t)b * a.M + m0 + r) * a.out_L + t;
    if (VEC && t + 3 < a.Q) {
      f32x4 v{acc[0], acc[1], acc[2], acc[3]};
      if (a.mask) {
        const f32x4 mk = *reinterpret_cast<const f32x4*>(a.mask + o);
        v.x *= mk.x > 0.f ? 1.f : mslope; v.y *= mk.y > 0.f ? 1.f : mslope;
        v.z *= mk.z > 0.f ? 1.f : mslope; v.w *= mk.w > 0.f ? 1.f : mslope;
      }
      if (a.res) {
        const f32x4 rs = *reinterpret_cast<const f32x4*>(a.res + o);
        v.x += rs.x; v.y += rs.y; v.z += rs.z; v.w += rs.w;
      }
      v.x = thin_act(v.x * a.out_scale, a.act, a.act_slope); v.y = thin_act(v.y * a.out_scale, a.act, a.act_slope);
      v.z = thin_act(v.z * a.out_scale, a.act, a.act_slope); v.w = thin_act(v.w * a.out_scale, a.act, a.act_slope);
      *reinterpret_cast<f32x4*>(a.out + o) = v;
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (t + PS * i >= a.Q) break;
        float v = acc[i];
        if (a.mask) v *= (a.mask[o + PS * i] > 0.f ? 1.f : mslope);
        if (a.res) v += a.res[o + PS * i];
        a.out[o + PS * i] = thin_act(v * a.out_scale, a.act, a.act_slope);
      }
    }
  }
}

// ---- one input channel, any row length: the [M, Q] outputs of a clip are one contiguous run; block = (clip, chunk of
// PASSES x 1024 flat outputs), thread = 4 consecutive flat outputs per pass (16-byte stores when the run allows it), each
// output looks up its row's taps and its K inputs in LDS (the clip's whole input row is staged once per block).
// PASSES: passes of a block (a chunk = PASSES * 1024 outputs)
template <int PASSES>
__global__ __launch_bounds__(RTG_THREADS) void cin1_flat_kernel(const ThinArgs a, const TileGeo g) {
  constexpr int kFlatChunk = PASSES * 4 * RTG_THREADS;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* wl = sm;                                     // [rows of this chunk][K]
  float* bl = wl + g.RB * a.K;                        // [rows of this chunk]
  float* xs = bl + g.RB;                              // [W], W = (Q-1)*stride + (K-1)*dil + 1 from position -pad
  const int tid = threadIdx.x;
  const int b = blockIdx.x / g.chunks, chunk = blockIdx.x - b * g.chunks;
  const int n_out = a.M * a.Q;
  const int e_lo = chunk * kFlatChunk, e_hi = min(n_out, e_lo + kFlatChunk);
  const int m_lo = e_lo / a.Q, rows = (e_hi - 1) / a.Q - m_lo + 1;
  for (int e = tid; e < rows * a.K; e += RTG_THREADS) {
    const int r = e / a.K, j = e - r * a.K;
    wl[e] = a.wp[packed_index(m_lo + r, 0, j, 1, a.K, a.tile_m, a.tap_major)];
  }
  for (int r = tid; r < rows; r += RTG_THREADS) bl[r] = a.bias ? a.bias[m_lo + r] : 0.f;
  cin1_stage(a, b, -a.pad, g.W, xs);
  __syncthreads();
  const size_t base = (size_t)b * n_out;
  for (int it = 0; it < kFlatChunk / (4 * RTG_THREADS); ++it) {
    const int e0 = e_lo + it * 4 * RTG_THREADS + 4 * tid;
    if (e0 >= n_out) break;
    float v[4];
    int mrow = e0 / a.Q, t = e0 - mrow * a.Q;
    mrow -= m_lo;
    const bool full = g.vec && e0 + 3 < n_out;
    float mk[4] = {1.f, 1.f, 1.f, 1.f}, rs[4] = {0.f, 0.f, 0.f, 0.f};
    if (full) {
      if (a.mask) { const f32x4 t4 = *reinterpret_cast<const f32x4*>(a.mask + base + e0); mk[0] = t4.x; mk[1] = t4.y; mk[2] = t4.z; mk[3] = t4.w; }
      if (a.res) { const f32x4 t4 = *reinterpret_cast<const f32x4*>(a.res + base + e0); rs[0] = t4.x; rs[1] = t4.y; rs[2] = t4.z; rs[3] = t4.w; }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (e0 + i < n_out && a.mask) mk[i] = a.mask[base + e0 + i];
        if (e0 + i < n_out && a.res) rs[i] = a.res[base + e0 + i];
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float s_ = 0.f;
      if (e0 + i < n_out) {
        s_ = bl[mrow];
        const float* wr = wl + mrow * a.K;
        const float* xp = xs + t * a.stride;
        for (int j = 0; j < a.K; ++j) s_ = fmaf(wr[j], xp[j * a.dil], s_);
        if (a.mask) s_ *= (mk[i] > 0.f ? 1.f : a.mask_slope);
        s_ += rs[i];
        s_ = thin_act(s_ * a.out_scale, a.act, a.act_slope);
      }
      v[i] = s_;
      if (++t == a.Q) { t = 0; ++mrow; }
    }
    if (g.vec && e0 + 3 < n_out) {
      *reinterpret_cast<f32x4*>(a.out + base + e0) = f32x4{v[0], v[1], v[2], v[3]};
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (e0 + i < n_out) a.out[base + e0 + i] = v[i];
    }
  }
}

}  // namespace

// rtg_thin2d.hip: the first Conv2d of StftDiscriminator (2 input channels)
bool rtg_thin2d_fwd_ok(const RtgConv1dDesc* d);
int rtg_thin2d_fwd_launch(const RtgConv1dDesc* d, const float* x, const float* wp, const float* bias, const float* mask,
                          const float* res, float* out, hipStream_t s);
bool rtg_thin2d_dgrad_ok(const RtgConv1dDesc* d);
int rtg_thin2d_dgrad_launch(const RtgConv1dDesc* d, const float* dy, const float* wp, const float* mask, const float* res,
                            float* out, hipStream_t s);

// 0: not a thin shape (use the MFMA kernel), 1: one input channel, 2: one output channel, 3: one output channel 3 x 3,
// 4 / 5: two input channels 3 x 3, forward / backward-data (rtg_thin2d.hip)
int rtg_thin_kind(const RtgConv1dDesc* d) {
  // (the two-channel layer along the frequency axis: its backward-data is a polyphase operator, shuf_S = 2)
  if (d->groups == 1 && d->C2 == 0 && d->shuf_S == 2 && (d->h_k > 1 || d->h_n > 1) && rtg_thin2d_dgrad_ok(d)) return 5;
  if (d->groups != 1 || d->C2 != 0 || d->shuf_S != 1 || d->out_split != 0 || d->accumulate) return 0;
  if (d->h_k > 1 || d->h_n > 1) {
    if (rtg_thin2d_fwd_ok(d)) return 4;
    if (rtg_thin2d_dgrad_ok(d)) return 5;
    // kind 3: one output channel, 3 x 3, stride 1, "same" padding, forward addressing (cout1_k3x3_kernel)
    if (d->Mg == 1 && d->h_mode == 0 && d->h_k == 3 && d->K == 3 && d->stride == 1 && d->h_stride == 1 && d->dil == 1 &&
        d->pad == 1 && d->h_pad == 1 && d->h_n == d->h_in && d->Q == d->L_in && d->out_L == d->Q && d->h_n > 0 &&
        d->B % d->h_n == 0 && d->Cg % 3 == 0 && (d->Cg / 3) % (8 * 8) == 0 && d->Cg / 3 <= 512 && !d->bf16 &&
        d->pre_mode < RTG_PRE_MUL_DLRELU && (long long)d->B * d->Cg / 3 * d->L_in * 4 < (1ll << 31) &&
        true)
      return 3;
    return 0;
  }
  if (d->Cg == 1 && d->K <= kMaxTaps) return 1;
  if (d->Mg == 1 && (long long)d->Cg * d->K <= 12288 && (long long)d->B * d->Cg * d->L_in * 4 < (1ll << 31)) return 2;
  return 0;
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
static inline int round4(int v) { return (v + 3) & ~3; }
constexpr int kLdsFloats = 12288;    // 48 KB per block: three blocks per CU overlap their load and compute phases

static inline int pow2_at_least(int v, int lo, int hi) {
  int p = lo;
  while (p < v && p < hi) p <<= 1;
  return p;
}
static inline int log2i(int p) {
  int s = 0;
  while ((1 << s) < p) ++s;
  return s;
}

// geometry of cout1_long_kernel for descriptor d; false: not served (fallback kernel)
static bool cout1_long_geo(const RtgConv1dDesc* d, bool vec, TileGeo* g) {
  const int Q = d->Q, B = d->B;
  if (d->K > kMaxTaps) return false;
  g->PT = 256;
  for (int pt = 512; pt >= 256; pt >>= 1)
    if ((long long)B * rtg_ceil_div(Q, pt) >= 768) { g->PT = pt; break; }
  g->R = 4; g->TP = g->PT / 4; g->NG = RTG_THREADS / g->TP;
  g->tp_shift = log2i(g->TP);
  g->unit = (d->stride == 1 && d->dil == 1) ? 1 : 0;
  g->shift = (vec && d->stride == 1) ? ((4 - (d->pad & 3)) & 3) : 0;
  g->tiles = rtg_ceil_div(Q + g->shift, g->PT);
  g->W = (g->PT - 1) * d->stride + (d->K - 1) * d->dil + 1;
  // pitch: the window plus what the aligned 16-byte reads of the last lane touch (4 * (TP - 1) + 20 floats), a multiple of 4
  const int need = g->unit ? 4 * (g->TP - 1) + 20 : g->W;
  g->Wp = round4((need > g->W ? need : g->W) + 4);
  const int Kp = round4(d->K);
  g->w_floats = round4(d->Cg * Kp);
  const int budget = kLdsFloats - g->w_floats;
  if (budget < g->NG * g->PT) return false;
  int ch = budget / g->Wp;
  if (ch > d->Cg) ch = d->Cg;
  ch -= ch % g->NG;
  if (ch < g->NG) return false;
  g->CH = ch;
  const int nq = vec ? (g->W + 3 + 3) / 4 : g->W;
  g->TPRow = pow2_at_least(nq, 32, RTG_THREADS);
  g->tprow_shift = log2i(g->TPRow);
  return true;
}

static bool cout1_short_geo(const RtgConv1dDesc* d, TileGeo* g) {
  if (d->Q > 256 || d->L_in > 1024) return false;
  g->TP = pow2_at_least(d->Q, 64, 256);
  g->tp_shift = log2i(g->TP);
  g->NG = kShortThreads / g->TP;
  g->R = 1; g->PT = g->TP; g->tiles = 1;
  g->w_floats = round4(d->Cg * d->K);
  int budget = kLdsFloats - g->w_floats - 4;
  if (budget > 4 * kShortRegs * kShortThreads) budget = 4 * kShortRegs * kShortThreads;   // what the prefetch holds
  if (budget < kShortThreads) return false;
  int ch = budget / d->L_in;
  if (ch > d->Cg) ch = d->Cg;
  const int mult = g->NG > 4 ? g->NG : 4;           // a multiple of 4 keeps every chunk 16-byte aligned (VEC)
  if (ch < d->Cg) ch -= ch % mult;
  if (ch < 1) return false;
  g->CH = ch;
  return true;
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
  const bool legacy = false;                                // (true: the round-1 kernels, kept as the general fallback)
  TileGeo g = {};
  if (kind == 4) return rtg_thin2d_fwd_launch(d, x, wp, bias, mask, res, out, s);
  if (kind == 5) return rtg_thin2d_dgrad_launch(d, x, wp, mask, res, out, s);
  if (kind == 3) {
    const int items = d->B / d->h_n, C = d->Cg / 3;
    const long long cols = (long long)items * d->h_in * d->L_in;
    RTG_KLAUNCH(cout1_k3x3_kernel, dim3((unsigned)rtg_ceil_div(cols, 64)), dim3(64 * kK3Waves), 0, s, a, C, d->h_in, items);
    return rtg_launch_status();
  }
  if (kind == 1 && !legacy && d->Q == d->out_L) {
    const long long wq = (long long)(d->Q - 1) * d->stride + (long long)(d->K - 1) * d->dil + 1;
    const bool vec_io = d->out_L % 4 == 0 && aligned16(out) && (!mask || aligned16(mask)) && (!res || aligned16(res));
    // (unaligned rows from 192 positions: a quarter-filled tile still beats cin1_flat_kernel's whole-row staging — period 11)
    if ((vec_io && d->Q >= 256) || (!vec_io && d->Q >= 192)) {
      g.tiles = rtg_ceil_div(d->Q, kRowTile);
      g.W = (kRowTile - 1) * d->stride + (d->K - 1) * d->dil + 1;
      const long long bx = (long long)d->B * g.tiles;
      int rb = d->Mg;
      if (bx < 1024) {
        const int want = rtg_ceil_div(1024, bx);
        rb = rtg_ceil_div(d->Mg, want);
        if (rb < 8) rb = 8;
        if (rb > d->Mg) rb = d->Mg;
      }
      g.RB = rb;
      const size_t lds = ((size_t)rb * (kMaxTaps + 1) + g.W) * sizeof(float);
      const int gy = rtg_ceil_div(d->Mg, rb);
      if (bx <= 0x7fffffffLL && gy <= 65535 && lds <= 48 * 1024) {
#define RTG_C1R(KT)                                                                                             \
  if (vec_io) RTG_KLAUNCH((cin1_rows_kernel<KT, true>), dim3((unsigned)bx, gy), dim3(RTG_THREADS), lds, s, a, g); \
  else RTG_KLAUNCH((cin1_rows_kernel<KT, false>), dim3((unsigned)bx, gy), dim3(RTG_THREADS), lds, s, a, g)
        switch (d->K) {
          case 3: RTG_C1R(3); break;
          case 5: RTG_C1R(5); break;
          case 7: RTG_C1R(7); break;
          case 15: RTG_C1R(15); break;
          default: RTG_C1R(0);
        }
#undef RTG_C1R
        return rtg_launch_status();
      }
    }
    const long long n_out = (long long)d->Mg * d->Q;
    // two passes per block (2048 outputs): conv_post backward-data (512 rows of 10-128 positions per clip) is a chain of
    // dependent mask loads and stores per thread — 4 x more blocks hide it better than 8 passes in one (measured in the
    // step, 14 launches: 0.44 -> 0.27 ms; one pass: the same)
    const int passes = 2;
    const int kFlatChunk = passes * 4 * RTG_THREADS;
    int rows_max = kFlatChunk / d->Q + 2;              // output rows a chunk can touch
    if (rows_max > d->Mg) rows_max = d->Mg;
    const size_t lds = ((size_t)rows_max * (d->K + 1) + wq) * sizeof(float);
    if (lds <= 48 * 1024 && n_out < (1ll << 30)) {
      g.W = (int)wq;
      g.RB = rows_max;
      g.chunks = rtg_ceil_div(n_out, kFlatChunk);
      g.vec = (n_out % 4 == 0 && aligned16(out) && (!mask || aligned16(mask)) && (!res || aligned16(res))) ? 1 : 0;
      const long long bx = (long long)d->B * g.chunks;
      if (bx <= 0x7fffffffLL) {
        if (passes == 8) RTG_KLAUNCH(cin1_flat_kernel<8>, dim3((unsigned)bx), dim3(RTG_THREADS), lds, s, a, g);
        else if (passes == 4) RTG_KLAUNCH(cin1_flat_kernel<4>, dim3((unsigned)bx), dim3(RTG_THREADS), lds, s, a, g);
        else if (passes == 2) RTG_KLAUNCH(cin1_flat_kernel<2>, dim3((unsigned)bx), dim3(RTG_THREADS), lds, s, a, g);
        else if (passes == 1) RTG_KLAUNCH(cin1_flat_kernel<1>, dim3((unsigned)bx), dim3(RTG_THREADS), lds, s, a, g);
        else return RTG_EINVAL;
        return rtg_launch_status();
      }
    }
  }
  if (kind == 2 && !legacy && !a.aux && d->K == 3 && d->stride == 1 && d->dil == 1 && d->pad == 1 && d->Q == d->L_in &&
      d->Q == d->out_L && d->Cg % (8 * 16) == 0 && d->Cg <= 1024 && (long long)d->B * d->Cg * d->L_in * 4 < (1ll << 31) &&
      true) {
    if (d->L_in <= 64 && d->L_in >= 4) {
      RTG_KLAUNCH(cout1_k3_rows_kernel, dim3((unsigned)d->B), dim3(64 * kK3Waves), 0, s, a);
      return rtg_launch_status();
    }
    const int cpb = d->L_in <= 64 ? 64 / d->L_in : 1;
    const int bpc = d->L_in <= 64 ? 1 : rtg_ceil_div(d->L_in, 64);
    const long long bx = bpc == 1 ? rtg_ceil_div(d->B, cpb) : (long long)d->B * bpc;
    if (bx <= 0x7fffffffLL) {
      RTG_KLAUNCH(cout1_k3_kernel, dim3((unsigned)bx), dim3(64 * kK3Waves), 0, s, a, cpb, bpc);
      return rtg_launch_status();
    }
  }
  if (kind == 2 && !legacy && !a.aux) {
    const bool short_ok = d->Q <= 256 && d->Q == d->out_L;
    const bool vec_s = ((long long)d->Cg * d->L_in) % 4 == 0 && aligned16(x);
    if (short_ok && cout1_short_geo(d, &g)) {
      size_t body = round4(g.CH * d->L_in) + 4;
      if (body < (size_t)kShortThreads) body = kShortThreads;
      const size_t lds = (g.w_floats + body) * sizeof(float);
      const bool v = vec_s && (g.CH % 4 == 0 || g.CH == d->Cg);
      if (v) RTG_KLAUNCH(cout1_short_kernel<true>, dim3((unsigned)d->B), dim3(kShortThreads), lds, s, a, g);
      else RTG_KLAUNCH(cout1_short_kernel<false>, dim3((unsigned)d->B), dim3(kShortThreads), lds, s, a, g);
      return rtg_launch_status();
    }
    const bool vec_l = d->L_in % 4 == 0 && aligned16(x);
    if (cout1_long_geo(d, vec_l, &g)) {
      const long long bx = (long long)d->B * g.tiles;
      const size_t body = (size_t)g.CH * g.Wp > (size_t)g.NG * g.PT ? (size_t)g.CH * g.Wp : (size_t)g.NG * g.PT;
      const size_t lds = (g.w_floats + body) * sizeof(float);
      if (bx <= 0x7fffffffLL) {
#define RTG_C1L(V, KT) RTG_KLAUNCH((cout1_long_kernel<V, KT>), dim3((unsigned)bx), dim3(RTG_THREADS), lds, s, a, g)
        if (vec_l) {
          switch (d->K) {
            case 3: RTG_C1L(true, 3); break;
            case 7: RTG_C1L(true, 7); break;
            case 15: RTG_C1L(true, 15); break;
            default: RTG_C1L(true, 0);
          }
        } else {
          RTG_C1L(false, 0);
        }
#undef RTG_C1L
        return rtg_launch_status();
      }
    }
  }
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
