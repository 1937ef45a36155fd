// rtg_wgrad.hip — convolution backward w.r.t. weight and bias on the fp32 matrix cores (see include/rtg.h).
//
// GEMM view: rows = output channels m of one group, columns = (input channel c, tap j) pairs of a channel chunk
// (+ one virtual "ones" column whose accumulator is the bias gradient), reduction over (clip, output position).
//   A[m][v]   = gy[b, m, t]                      staged as an LDS tile [rows][TT] with an odd row pitch
//   B[v][c,j] = pre(x[b, c, t*stride - pad + j*dil])   read from an LDS patch [channels][ROW]; the lane's (c,j) fixes a
//               base address once per tile, the reduction loop only adds stride per step
// The reduction index v ("virtual position") walks tiles of TT = 64 positions.  Long rows: a tile is 64 consecutive
// positions of one clip.  Short rows (MPD / MSD tails, 10..60 positions): several clips are packed side by side in one
// tile (segments of seg_len positions, pitch seg_len*stride in the patch) so the MFMA reduction is not spent on padding.
// A block owns (group, a band of output rows, one channel chunk, one split of the reduction).  Its 4 waves form a
// WM x WN grid and each wave keeps an MTW x NTW register tile of MFMA accumulators, so one A fragment feeds NTW and one
// B fragment feeds MTW MFMAs (the 2x2 form halves the LDS reads per MFMA).  The next tile's operands are fetched into
// registers by bounds-checked buffer loads while the current tile is multiplied; LDS is single-buffered (two barriers
// per tile) so that two workgroups fit a CU and cover each other's barrier phases.  Partials are stored per split
// (fixed-order reduction in rtg_weightnorm_backward).
#include <cstdlib>

#include "rtg_common.h"

namespace {

constexpr int TT = 64;          // reduction (virtual position) steps per staged tile
constexpr int ROWD = 81;        // LDS pitch of the gy tile: odd, and 81^-1 = 17 (mod 32) keeps 16-row reads conflict free

using rsrc_t = __amdgpu_buffer_rsrc_t;
#define RTG_OOB 0x80000000u

__device__ __forceinline__ float buf_load(rsrc_t r, unsigned off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
}

struct WgArgs {
  const float *x1, *x2, *dy, *gy_aux;
  float* part;
  int B, C1, C2, L_in, groups, Cg, Mg, K, stride, dil, pad, Q, dy_L;
  int pre_mode;
  float pre_slope;
  int gy_mode;
  float gy_slope, gy_scale;
  int splits;
  long long part_stride;
  int CKW, n_cchunk, m_blocks, n_ttiles, n_tiles_total, PW, ROW, ones_off, xbuf_sz;
  int cont;                                 // 1: one virtual sequence over all clips, 0: tiles never cross clips
  int seg_len, seg_pitch, seg_pw;           // virtual positions per clip, its pitch in the patch, its patch width
  float inv_seg, inv_pitch;
  int two_d, h_in, h_k, h_stride, h_pad, h_n;   // second dimension, see RtgConv1dDesc
  int x_bytes, dy_bytes;
};

template <int TM>
struct MfmaW;
template <>
struct MfmaW<32> {
  using acc_t = f32x16;
  static constexpr int NREG = 16;
  static __device__ __forceinline__ acc_t run(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int row(int lane, int r) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }
};
template <>
struct MfmaW<16> {
  using acc_t = f32x4;
  static constexpr int NREG = 4;
  static __device__ __forceinline__ acc_t run(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int row(int lane, int r) { return (lane >> 4) * 4 + r; }
};

// block shapes: wave grid WM x WN (WM*WN = 4), register tile MTW x NTW
template <int TM, int MTW, int NTW, int WM, int MAXIT>
__global__ __launch_bounds__(RTG_THREADS) void wgrad_kernel(const WgArgs a) {
  using M = MfmaW<TM>;
  using acc_t = typename M::acc_t;
  constexpr int WN = 4 / WM;
  constexpr int KK = 64 / TM;
  constexpr int ROWS = WM * MTW * TM;          // gy rows of the block
  constexpr int DR = ROWS / 4;                 // gy rows staged per wave
  constexpr int XR = (MAXIT <= 4) ? 8 : 4;     // patch rows staged per wave (CKW <= 4*XR)
  constexpr int GRP = 8;                       // k-steps per read phase

  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave - wm * WN;

  int by = blockIdx.y;
  const int cchunk = by % a.n_cchunk; by /= a.n_cchunk;
  const int mb = by % a.m_blocks;
  const int g = by / a.m_blocks;
  const int split = blockIdx.x;
  const int c0 = cchunk * a.CKW;
  const int cw = min(a.CKW, a.Cg - c0);
  const int m0 = mb * ROWS;                          // first row (within group) of this block

  float* xb = lds;
  float* db = lds + a.xbuf_sz;
  float* ones = lds + a.ones_off;
  for (int i = tid; i < TT * a.stride + 8; i += RTG_THREADS) ones[i] = 1.f;

  // ---- accumulators and LDS operand bases of this wave's register tile
  acc_t acc[MTW][NTW];
  int a_base[MTW], b_base[NTW];
  const int n_lane = lane & (TM - 1), kk = lane / TM;
#pragma unroll
  for (int i = 0; i < MTW; ++i) {
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
      for (int r = 0; r < M::NREG; ++r) acc[i][j][r] = 0.f;
    a_base[i] = ((wm * MTW + i) * TM + n_lane) * ROWD + kk;
  }
#pragma unroll
  for (int j = 0; j < NTW; ++j) {
    const int n = (wn * NTW + j) * TM + n_lane;      // column within the chunk
    const int cl = n / a.K, jj = n - cl * a.K;
    b_base[j] = (cl < a.CKW) ? (cl * a.ROW + jj * a.dil + kk * a.stride) : (-(1 << 20) + kk * a.stride);
  }

  // n / d for 0 <= n < 2^24 through the float reciprocal, exact after one correction step either way
  auto fdiv = [](int n, int d, float inv, int& rem) __attribute__((always_inline)) {
    int q = (int)((float)n * inv);
    int r = n - q * d;
    if (r < 0) { --q; r += d; }
    else if (r >= d) { ++q; r -= d; }
    rem = r;
    return q;
  };
  int xseg[MAXIT], xw[MAXIT];

  const rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc((void*)a.x1, 0, a.x_bytes, 0x00020000);
  const rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x2 ? a.x2 : a.x1), 0,
                                                      a.x2 ? a.B * a.C2 * a.L_in * 4 : 0, 0x00020000);
  const int dy_bytes = a.dy_bytes;
  const rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, dy_bytes, 0x00020000);
  const rsrc_t raux = __builtin_amdgcn_make_buffer_rsrc((void*)(a.gy_aux ? a.gy_aux : a.dy), 0,
                                                        a.gy_aux ? dy_bytes : 0, 0x00020000);
  const bool has_aux = a.gy_aux != nullptr && (a.gy_mode == RTG_PRE_MUL_DLRELU || a.gy_mode == RTG_PRE_MUL_DTANH);
  const float xslope = (a.pre_mode == RTG_PRE_LRELU) ? a.pre_slope : 1.f;
  const float gslope = (a.gy_mode == RTG_PRE_LRELU) ? a.gy_slope : 1.f;

  float sx[XR][MAXIT], sd[DR], sa[DR];

  auto gload = [&](int tl) __attribute__((always_inline)) {
    // the reduction walks ONE virtual sequence: clip c owns positions [c*seg_len, c*seg_len + Q) (the rest of its
    // seg_len slots is a gap with gy = 0, wide enough that the next clip's input patch does not overlap); tile tl
    // covers virtual positions [tl*TT, tl*TT + TT) whatever clip boundaries fall inside it
    // (long rows whose length fills whole tiles keep the cheaper per-clip tiling: tile = (clip, 64-step window))
    int b0 = 0, o_start = -a.pad, t0 = 0, dseg = 0, dt = lane;
    if (a.cont) {
      const int v0 = tl * TT;
#pragma unroll
      for (int it = 0; it < MAXIT; ++it) {
        const int o = lane + 64 * it;
        int w;
        xseg[it] = fdiv(v0 * a.stride + o, a.seg_pitch, a.inv_pitch, w);
        xw[it] = (o < a.PW && w < a.seg_pw) ? w : -(1 << 28);             // never valid
      }
      dseg = fdiv(v0 + lane, a.seg_len, a.inv_seg, dt);
    } else {
      b0 = tl / a.n_ttiles;
      t0 = (tl - b0 * a.n_ttiles) * TT;
      o_start = t0 * a.stride - a.pad;
#pragma unroll
      for (int it = 0; it < MAXIT; ++it) {
        xseg[it] = 0;
        xw[it] = (lane + 64 * it < a.PW) ? lane + 64 * it : -(1 << 28);
      }
    }
    const bool dcol_ok = true;
    if (a.two_d) {
      // channel (ci, kh) of clip (item, row r) reads input row r*h_stride - h_pad + kh of [items, C, h_in, L_in]
      const int cin = a.C1 / a.h_k;
      const unsigned item_bytes = (unsigned)cin * (unsigned)a.h_in * (unsigned)a.L_in * 4u;
#pragma unroll
      for (int i = 0; i < XR; ++i) {
        const int cl = wave * XR + i;
        const int c = c0 + cl;
        const int ci = c / a.h_k, kh = c - ci * a.h_k;
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
          const int pos = o_start + xw[it];
          const int bb = b0 + xseg[it];
          const int item = bb / a.h_n, hh = bb - item * a.h_n;
          const int hrow = hh * a.h_stride - a.h_pad + kh;
          const bool ok = cl < cw && pos >= 0 && pos < a.L_in && bb < a.B && hrow >= 0 && hrow < a.h_in;
          const unsigned off = ok ? (unsigned)item * item_bytes +
                                        ((unsigned)(ci * a.h_in + hrow) * (unsigned)a.L_in + (unsigned)pos) * 4u
                                  : RTG_OOB;
          sx[i][it] = buf_load(r1, off);
        }
      }
    } else
#pragma unroll
    for (int i = 0; i < XR; ++i) {
      const int cl = wave * XR + i;
      const int gc = g * a.Cg + c0 + cl;
      const bool in1 = gc < a.C1;
      const rsrc_t r = in1 ? r1 : r2;
      const unsigned cstride = (unsigned)(in1 ? a.C1 : a.C2) * (unsigned)a.L_in * 4u;
      const unsigned rowoff = (unsigned)(in1 ? gc : gc - a.C1) * (unsigned)a.L_in * 4u;
      const unsigned rowoob = (cl < cw) ? 0u : RTG_OOB;
#pragma unroll
      for (int it = 0; it < MAXIT; ++it) {
        const int pos = o_start + xw[it];
        const int bb = b0 + xseg[it];
        const bool ok = pos >= 0 && pos < a.L_in && bb < a.B;   // xw = -2^28 makes pos negative
        const unsigned off = ok ? ((unsigned)bb * cstride + rowoff + (unsigned)pos * 4u) | rowoob : RTG_OOB;
        sx[i][it] = buf_load(r, off);
      }
    }
    {
      const int bb = b0 + dseg;
      const int t = t0 + dt;
      const bool colok = dcol_ok && bb < a.B && t < a.Q;
      // dy is [items, rows, h_n, dy_L]: element (item, m, hh, t); h_n == 1 in 1-D
      const int item = bb / a.h_n, hh = bb - item * a.h_n;
      const unsigned rowpitch = (unsigned)a.h_n * (unsigned)a.dy_L * 4u;
      const unsigned coloff = colok ? (unsigned)item * (unsigned)(a.groups * a.Mg) * rowpitch +
                                          ((unsigned)hh * (unsigned)a.dy_L + (unsigned)t) * 4u
                                    : RTG_OOB;
#pragma unroll
      for (int i = 0; i < DR; ++i) {
        const int rl = wave + 4 * i;
        const int m = m0 + rl;
        const unsigned off = (m < a.Mg) ? (coloff + (unsigned)(g * a.Mg + m) * rowpitch) | (coloff & RTG_OOB)
                                        : RTG_OOB;
        sd[i] = buf_load(rdy, off);
        if (has_aux) sa[i] = buf_load(raux, off);
      }
    }
  };
  auto swrite = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < XR; ++i) {
      const int cl = wave * XR + i;
      if (cl < a.CKW) {
#pragma unroll
        for (int it = 0; it < MAXIT; ++it)
          if (lane + 64 * it < a.PW) {
            float v = sx[i][it];
            asm volatile("" : "+v"(v) : : "memory");   // consume the prefetched value here, below the MFMA loop
            xb[cl * a.ROW + lane + 64 * it] = v > 0.f ? v : v * xslope;
          }
      }
    }
#pragma unroll
    for (int i = 0; i < DR; ++i) {
      const int rl = wave + 4 * i;
      float v = sd[i];
      asm volatile("" : "+v"(v) : : "memory");
      if (has_aux) {
        const float av = sa[i];
        v *= (a.gy_mode == RTG_PRE_MUL_DTANH) ? (1.f - av * av) : (av > 0.f ? 1.f : a.gy_slope);
      } else {
        v = v > 0.f ? v : v * gslope;
      }
      db[rl * ROWD + lane] = v * a.gy_scale;
    }
  };

  const int total = a.n_tiles_total;
  int tl = split;
  if (tl < total) {
    gload(tl);
    swrite();
  }
  __syncthreads();
  for (; tl < total; tl += a.splits) {
    const bool more = tl + a.splits < total;
    if (more) gload(tl + a.splits);
    const float* ap[MTW];
    const float* bp[NTW];
#pragma unroll
    for (int i = 0; i < MTW; ++i) ap[i] = db + a_base[i];
#pragma unroll
    for (int j = 0; j < NTW; ++j) bp[j] = (b_base[j] >= 0) ? xb + b_base[j] : ones + (b_base[j] + (1 << 20));
#pragma unroll
    for (int t0 = 0; t0 < TT; t0 += GRP * KK) {
      float af[MTW][GRP], bf[NTW][GRP];
#pragma unroll
      for (int u = 0; u < GRP; ++u) {
#pragma unroll
        for (int i = 0; i < MTW; ++i) af[i][u] = ap[i][t0 + u * KK];
#pragma unroll
        for (int j = 0; j < NTW; ++j) bf[j][u] = bp[j][(t0 + u * KK) * a.stride];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < GRP; ++u)
#pragma unroll
        for (int i = 0; i < MTW; ++i)
#pragma unroll
          for (int j = 0; j < NTW; ++j) acc[i][j] = M::run(af[i][u], bf[j][u], acc[i][j]);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();                 // every wave is done reading the tile
    if (more) swrite();
    __syncthreads();
  }

  // ---- store this split's partial
  float* wpart = a.part + (size_t)split * a.part_stride;
  float* bpart = wpart + (size_t)a.groups * a.Mg * a.Cg * a.K;
  const int nb = a.CKW * a.K;                        // the "ones" column
#pragma unroll
  for (int i = 0; i < MTW; ++i)
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
      const int n = (wn * NTW + j) * TM + n_lane;
#pragma unroll
      for (int r = 0; r < M::NREG; ++r) {
        const int m = m0 + (wm * MTW + i) * TM + M::row(lane, r);
        if (m >= a.Mg) continue;
        const size_t rowg = (size_t)g * a.Mg + m;
        if (n < cw * a.K) wpart[rowg * (a.Cg * a.K) + (size_t)c0 * a.K + n] = acc[i][j][r];
        else if (n == nb && cchunk == 0) bpart[rowg] = acc[i][j][r];
      }
    }
}

struct Shape {
  int MTW, NTW, WM;
};
// menu of block shapes (register tile, wave grid)
constexpr Shape kShapes[] = {{2, 2, 2}, {2, 2, 1}, {1, 2, 1}, {1, 4, 1}, {1, 1, 1}, {1, 1, 4}};
constexpr int kNumShapes = sizeof(kShapes) / sizeof(Shape);

struct WgGeom {
  int TM, shape, CKW, n_cchunk, m_blocks, n_ttiles, n_tiles_total, PW, ROW, maxit;
  int seg_len, seg_pw, cont;
};

int geometry(const RtgWgradDesc* d, WgGeom* o) {
  const int TM = d->Mg >= 32 ? 32 : 16;
  const int n_mt = rtg_ceil_div(d->Mg, TM);
  o->TM = TM;
  o->PW = (TT - 1) * d->stride + (d->K - 1) * d->dil + 1;
  if (o->PW > RTG_PW_MAX) return RTG_ERANGE;
  o->maxit = o->PW <= 128 ? 2 : (o->PW <= 256 ? 4 : RTG_PW_MAX / 64);
  const int ckw_cap = (o->maxit <= 4) ? 32 : 16;
  double best = -1.0;
  o->shape = -1;
  for (int s = 0; s < kNumShapes; ++s) {
    const Shape sh = kShapes[s];
    const int bm = sh.WM * sh.MTW, bn = (4 / sh.WM) * sh.NTW;     // block tile in MFMA tiles
    int ckw = (bn * TM - 1) / d->K;
    if (ckw > d->Cg) ckw = d->Cg;
    if (ckw > ckw_cap) ckw = ckw_cap;
    if (ckw < 1) continue;
    const int n_cchunk = rtg_ceil_div(d->Cg, ckw);
    const int m_blocks = rtg_ceil_div(n_mt, bm);
    const double eff_m = (double)d->Mg / ((double)m_blocks * bm * TM);
    const double eff_n = ((double)d->Cg * d->K + 1.0) / ((double)n_cchunk * bn * TM);
    const double reuse = (double)(sh.MTW * sh.NTW) / (sh.MTW + sh.NTW);
    const double score = eff_m * eff_n * (0.55 + 0.45 * (reuse > 1.0 ? 1.0 : reuse));
    if (score > best) {
      best = score;
      o->shape = s; o->CKW = ckw; o->n_cchunk = n_cchunk; o->m_blocks = m_blocks;
    }
  }
  if (o->shape < 0) return RTG_ERANGE;
  // one virtual sequence over all clips: seg_len slots per clip (its Q outputs + the gap that separates patches)
  const int extra = (d->K - 1) * d->dil + 1 - d->stride;
  const int Lseg = d->Q + (extra > 0 ? (extra + d->stride - 1) / d->stride : 0);
  o->seg_pw = (d->Q - 1) * d->stride + (d->K - 1) * d->dil + 1;
  o->seg_len = Lseg;
  if ((long long)d->B * Lseg * d->stride + RTG_PW_MAX >= (1ll << 24)) return RTG_ERANGE;   // float-reciprocal division
  const int per_clip = rtg_ceil_div(d->Q, TT);
  o->cont = ((double)d->Q / Lseg > 1.08 * (double)d->Q / ((double)per_clip * TT) && d->B >= 2) ? 1 : 0;
  if (const char* f = getenv("RTG_DEV_WGRAD_CONT")) o->cont = (f[0] == '1' && d->B >= 2) ? 1 : 0;   // tuning aid
  if (o->cont) {
    o->n_ttiles = 1;
    o->n_tiles_total = rtg_ceil_div((long long)d->B * Lseg, TT);
  } else {
    o->n_ttiles = per_clip;
    o->n_tiles_total = d->B * per_clip;
  }
  const int want = (d->K * d->dil) & 31;
  int row = o->PW;
  while ((row & 31) != want) ++row;
  o->ROW = row;
  return RTG_OK;
}

int validate(const RtgWgradDesc* d) {
  if (d->B < 1 || d->C1 < 1 || d->C2 < 0 || d->L_in < 1 || d->groups < 1 || d->Cg < 1 || d->Mg < 1 || d->K < 1 ||
      d->stride < 1 || d->dil < 1 || d->Q < 1 || d->dy_L < d->Q)
    return RTG_EINVAL;
  if (d->C1 + d->C2 != d->groups * d->Cg) return RTG_EINVAL;
  if (d->groups > 1 && d->C2 != 0) return RTG_EINVAL;
  if (d->stride > 8) return RTG_ERANGE;
  const bool two_d = d->h_k > 1 || d->h_n > 1;
  if (two_d) {
    if (d->h_in < 1 || d->h_k < 1 || d->h_stride < 1 || d->h_pad < 0 || d->h_n < 1) return RTG_EINVAL;
    if (d->groups != 1 || d->C2 != 0 || d->C1 % d->h_k != 0 || d->B % d->h_n != 0) return RTG_EINVAL;
  }
  // 32-bit buffer offsets
  const long long xb = two_d ? (long long)(d->B / d->h_n) * (d->C1 / d->h_k) * d->h_in * d->L_in * 4
                             : (long long)d->B * (d->C1 > d->C2 ? d->C1 : d->C2) * d->L_in * 4;
  if (xb >= (1ll << 31)) return RTG_ERANGE;
  if ((long long)d->B * d->groups * d->Mg * d->dy_L * 4 >= (1ll << 31)) return RTG_ERANGE;
  return RTG_OK;
}

template <int TM, int MTW, int NTW, int WM, int MAXIT>
int launch(const WgArgs& a, dim3 grid, size_t lds_bytes, hipStream_t s) {
  auto k = wgrad_kernel<TM, MTW, NTW, WM, MAXIT>;
  if (lds_bytes > 64 * 1024) hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  hipLaunchKernelGGL(k, grid, dim3(RTG_THREADS), lds_bytes, s, a);
  return rtg_launch_status();
}

template <int TM, int MTW, int NTW, int WM>
int launch_it(int maxit, const WgArgs& a, dim3 grid, size_t lds_bytes, hipStream_t s) {
  if (maxit <= 2) return launch<TM, MTW, NTW, WM, 2>(a, grid, lds_bytes, s);
  if (maxit <= 4) return launch<TM, MTW, NTW, WM, 4>(a, grid, lds_bytes, s);
  return launch<TM, MTW, NTW, WM, RTG_PW_MAX / 64>(a, grid, lds_bytes, s);
}

template <int TM>
int launch_shape(int shape, int maxit, const WgArgs& a, dim3 grid, size_t lds_bytes, hipStream_t s) {
  switch (shape) {
    case 0: return launch_it<TM, 2, 2, 2>(maxit, a, grid, lds_bytes, s);
    case 1: return launch_it<TM, 2, 2, 1>(maxit, a, grid, lds_bytes, s);
    case 2: return launch_it<TM, 1, 2, 1>(maxit, a, grid, lds_bytes, s);
    case 3: return launch_it<TM, 1, 4, 1>(maxit, a, grid, lds_bytes, s);
    case 4: return launch_it<TM, 1, 1, 1>(maxit, a, grid, lds_bytes, s);
    default: return launch_it<TM, 1, 1, 4>(maxit, a, grid, lds_bytes, s);
  }
}

}  // namespace

extern "C" int rtg_wgrad_splits(const RtgWgradDesc* d) {
  if (!d) return RTG_ENULL;
  int st = validate(d);
  if (st) return st;
  WgGeom g;
  st = geometry(d, &g);
  if (st) return st;
  const long long base = (long long)d->groups * g.m_blocks * g.n_cchunk;
  const long long total = g.n_tiles_total;
  long long s = (640 + base - 1) / base;         // aim at ~2.5 blocks per CU
  if (s > total) s = total;
  if (s > 512) s = 512;
  if (s < 1) s = 1;
  return (int)s;
}

extern "C" int rtg_conv1d_wgrad(const RtgWgradDesc* d, const float* x1, const float* x2, const float* dy,
                                const float* gy_aux, float* part, void* stream) {
  if (!d || !x1 || !dy || !part) return RTG_ENULL;
  int st = validate(d);
  if (st) return st;
  if (d->C2 > 0 && !x2) return RTG_ENULL;
  if ((d->gy_mode == RTG_PRE_MUL_DLRELU || d->gy_mode == RTG_PRE_MUL_DTANH) && !gy_aux) return RTG_ENULL;
  if (d->pre_mode != RTG_PRE_NONE && d->pre_mode != RTG_PRE_LRELU) return RTG_EINVAL;
  if (d->splits < 1 || d->splits > 65535) return RTG_EINVAL;
  const long long need = (long long)d->groups * d->Mg * ((long long)d->Cg * d->K + 1);
  if (d->splits > 1 && d->part_stride < need) return RTG_EINVAL;
  WgGeom g;
  st = geometry(d, &g);
  if (st) return st;
  const Shape sh = kShapes[g.shape];

  WgArgs a;
  a.x1 = x1; a.x2 = x2; a.dy = dy; a.gy_aux = gy_aux; a.part = part;
  a.B = d->B; a.C1 = d->C1; a.C2 = d->C2; a.L_in = d->L_in; a.groups = d->groups; a.Cg = d->Cg; a.Mg = d->Mg;
  a.K = d->K; a.stride = d->stride; a.dil = d->dil; a.pad = d->pad; a.Q = d->Q; a.dy_L = d->dy_L;
  a.pre_mode = d->pre_mode; a.pre_slope = d->pre_slope; a.gy_mode = d->gy_mode; a.gy_slope = d->gy_slope;
  a.gy_scale = d->gy_scale;
  a.splits = d->splits; a.part_stride = d->part_stride;
  a.CKW = g.CKW; a.n_cchunk = g.n_cchunk; a.m_blocks = g.m_blocks;
  a.n_ttiles = g.n_ttiles; a.n_tiles_total = g.n_tiles_total; a.PW = g.PW; a.ROW = g.ROW;
  a.seg_len = g.seg_len; a.seg_pw = g.seg_pw; a.cont = g.cont;
  a.seg_pitch = g.seg_len * d->stride;
  a.inv_seg = 1.0f / (float)a.seg_len; a.inv_pitch = 1.0f / (float)a.seg_pitch;
  const bool two_d = d->h_k > 1 || d->h_n > 1;
  a.two_d = two_d ? 1 : 0;
  a.h_in = two_d ? d->h_in : 1; a.h_k = two_d ? d->h_k : 1; a.h_stride = two_d ? d->h_stride : 1;
  a.h_pad = two_d ? d->h_pad : 0; a.h_n = two_d ? d->h_n : 1;
  a.x_bytes = two_d ? (d->B / d->h_n) * (d->C1 / d->h_k) * d->h_in * d->L_in * 4 : d->B * d->C1 * d->L_in * 4;
  a.dy_bytes = d->B * d->groups * d->Mg * d->dy_L * 4;
  const int rows = sh.WM * sh.MTW * g.TM;
  const int xr_cap = (g.maxit <= 4) ? 32 : 16;
  a.xbuf_sz = xr_cap * g.ROW;                       // patch rows up to the staging capacity (rows past CKW unused)
  if (g.CKW < xr_cap) a.xbuf_sz = g.CKW * g.ROW;
  a.ones_off = a.xbuf_sz + rows * ROWD;
  const size_t lds_bytes = (size_t)(a.ones_off + TT * d->stride + 8) * sizeof(float);
  const long long gy = (long long)d->groups * g.m_blocks * g.n_cchunk;
  if (gy > 65535) return RTG_ERANGE;
  dim3 grid(d->splits, (unsigned)gy, 1);
  hipStream_t s = (hipStream_t)stream;
  if (g.TM == 32) return launch_shape<32>(g.shape, g.maxit, a, grid, lds_bytes, s);
  return launch_shape<16>(g.shape, g.maxit, a, grid, lds_bytes, s);
}
