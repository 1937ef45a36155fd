// rtg_wgrad_kernel.h — device side of rtg_conv1d_wgrad (kernel + template dispatch), shared by the four translation
// units rtg_wgrad_m{0..3}.hip, one per addressing mode (bit 0: continuous virtual sequence, bit 1: 2-D rows), so that the
// mode is a compile-time constant in the staging code and the four sets of instances build in parallel.
// See rtg_wgrad.hip for the algorithm notes and the host side.
#pragma once
#include "rtg_common.h"

namespace rtg_wg {

constexpr int TT = 64;          // reduction (virtual position) steps per staged tile
constexpr int ROWD = 81;        // LDS pitch of the gy tile: odd, and 81^-1 = 17 (mod 32) keeps 16-row reads conflict free

using rsrc_t = __amdgpu_buffer_rsrc_t;
#define RTG_OOB 0x80000000u

__device__ __forceinline__ float buf_load(rsrc_t r, unsigned off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
}

#define RTG_WG_MAX_GROUP RTG_WGRAD_MAX_GROUP

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
  int gy, n_items, per_xcd;                 // block -> work mapping (XCD-aware, see the kernel)
};

typedef short bf4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bf4 to_bf4(float v0, float v1, float v2, float v3) {
  bf4 r;
  r.x = __builtin_bit_cast(short, (__bf16)v0);
  r.y = __builtin_bit_cast(short, (__bf16)v1);
  r.z = __builtin_bit_cast(short, (__bf16)v2);
  r.w = __builtin_bit_cast(short, (__bf16)v3);
  return r;
}

template <int TM>
struct MfmaW;
template <>
struct MfmaW<32> {
  using acc_t = f32x16;
  static constexpr int NREG = 16;
  static __device__ __forceinline__ acc_t run(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ acc_t runbf(bf4 a, bf4 b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a, b, c, 0, 0, 0);
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
  static __device__ __forceinline__ acc_t runbf(bf4 a, bf4 b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int row(int lane, int r) { return (lane >> 4) * 4 + r; }
};

// block shapes: wave grid WM x WN (WM*WN = 4), register tile MTW x NTW
// `bid`: the block's index within its problem (blockIdx.x of a plain launch; a grouped launch subtracts the first block of
// the member, see wgrad_group_kernel)
template <int TM, int MTW, int NTW, int WM, int MAXIT, int MODE>
__device__ __forceinline__ void wgrad_body(const WgArgs& a, const unsigned bid) {
  using M = MfmaW<TM>;
  constexpr bool CONT = (MODE & 1) != 0, TWO_D = (MODE & 2) != 0;   // compile-time addressing mode
  // bit 2: bf16 operands — the fp32 tiles in LDS are rounded to bf16 as they are read into fragments (a lane holds 4
  // consecutive reduction positions), multiplied on the bf16 matrix cores, accumulated in fp32
  constexpr bool BF = (MODE & 4) != 0;
  constexpr int KL = BF ? 4 : 1;                // consecutive reduction positions per lane and MFMA
  using acc_t = typename M::acc_t;
  constexpr int WN = 4 / WM;
  constexpr int KK = 64 / TM;
  constexpr int ROWS = WM * MTW * TM;          // gy rows of the block
  constexpr int DR = ROWS / 4;                 // gy rows staged per wave
  constexpr int XR = (MAXIT <= 4) ? 8 : 4;     // patch rows staged per wave (CKW <= 4*XR)
  constexpr int GRP = 8;                       // k-steps per read phase

#ifdef RTG_EXP_EMPTY
  return;
#endif
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave - wm * WN;

  // Block -> work item, XCD-aware: workgroups are dealt round-robin over the 8 XCDs (block b runs on XCD b % 8), each
  // with its own L2.  Work items are ordered (split, group, row band, channel chunk): the blocks of one split read the
  // same positions of x and dy — the row bands share an x chunk, the channel chunks share a dy band.  XCD k takes the
  // contiguous item range [k * per_xcd, (k + 1) * per_xcd), so the blocks resident on an XCD at one time share their
  // operand tiles through that XCD's L2 (before: the split index ran fastest and neighbours shared nothing; PMC l2_hit
  // 0.06, 176-201 MB fetched per launch for 26 MB of operands).
  const int item = (int)(bid & 7u) * a.per_xcd + (int)(bid >> 3);
  if (item >= a.n_items) return;
  const int split = item / a.gy;
  int by = item - split * a.gy;
  const int cchunk = by % a.n_cchunk; by /= a.n_cchunk;
  const int mb = by % a.m_blocks;
  const int g = by / a.m_blocks;
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
    a_base[i] = ((wm * MTW + i) * TM + n_lane) * ROWD + kk * KL;
  }
#pragma unroll
  for (int j = 0; j < NTW; ++j) {
    const int n = (wn * NTW + j) * TM + n_lane;      // column within the chunk
    const int cl = n / a.K, jj = n - cl * a.K;
    b_base[j] = (cl < a.CKW) ? (cl * a.ROW + jj * a.dil + kk * KL * a.stride) : (-(1 << 20) + kk * KL * a.stride);
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

  // Every staged row (a channel of the patch, an output row of gy) is fetched through its OWN buffer descriptor held in
  // SGPRs: base = the row's first element, records = its valid bytes.  One per-lane byte offset per 64-column block then
  // serves all rows of the tile, the per-load address arithmetic is scalar, and the hardware range check supplies the
  // zero padding on the right edge (offsets left of the row are replaced by RTG_OOB).  The row bases form a chain
  // (next = previous + pitch) that is kept opaque to the optimiser: expanded into one hoisted base per row it would
  // spill the scalar register file.
  using u64 = unsigned long long;
  auto chain = [](u64& p) __attribute__((always_inline)) { asm volatile("" : "+s"(p)); };
  auto chain32 = [](int& v) __attribute__((always_inline)) { asm volatile("" : "+s"(v)); };
  auto desc = [](u64 p, int bytes) __attribute__((always_inline)) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, bytes < 0 ? 0 : bytes, 0x00020000);
  };
  const bool has_aux = a.gy_aux != nullptr && (a.gy_mode == RTG_PRE_MUL_DLRELU || a.gy_mode == RTG_PRE_MUL_DTANH);
  const float xslope = (a.pre_mode == RTG_PRE_LRELU) ? a.pre_slope : 1.f;
  const float gslope = (a.gy_mode == RTG_PRE_LRELU) ? a.gy_slope : 1.f;
  const int cin = a.C1 / a.h_k;                     // 2-D: real input channels
  const int x2_bytes = a.B * a.C2 * a.L_in * 4;
  const int rows_all = a.groups * a.Mg;
  const float inv_hn = 1.0f / (float)a.h_n;
  const u64 rowb = (u64)a.L_in * 4;                 // bytes of one input row
  const u64 dpitch = (u64)a.h_n * a.dy_L * 4;       // bytes between output rows m and m + 1
  const int c_first = c0 + wave * XR;               // first patch row this wave stages
  const int ci_first = TWO_D ? c_first / a.h_k : 0;
  const int kh_first = TWO_D ? c_first - ci_first * a.h_k : 0;
  const int m_first = m0 + wave;                    // first gy row this wave stages (then every 4th)

  float sx[XR][MAXIT], sd[DR], sa[DR];

  auto gload = [&](int tl) __attribute__((always_inline)) {
    // the reduction walks tiles of TT virtual positions.  Per-clip tiling: tile = (clip b0, 64-step window), everything
    // but the lane's column is uniform.  Continuous mode: clip c owns virtual positions [c*seg_len, c*seg_len + Q) (the
    // rest of its seg_len slots is a gap with gy = 0, wide enough that the next clip's patch does not overlap) and a
    // tile covers [tl*TT, tl*TT + TT) whatever clip boundaries fall inside, so the clip is a per-lane quantity.
    unsigned vx1[MAXIT], vx2[MAXIT], vd;
    int hb[MAXIT];                                  // continuous 2-D: the lane's input row for kernel row 0
    int b0 = 0, item0 = 0, hh0 = 0;
    if (CONT) {
      const int v0 = tl * TT;
#pragma unroll
      for (int it = 0; it < MAXIT; ++it) {
        const int o = lane + 64 * it;
        int w;
        const int seg = fdiv(v0 * a.stride + o, a.seg_pitch, a.inv_pitch, w);
        const int pos = w - a.pad;
        const bool ok = o < a.PW && w < a.seg_pw && pos >= 0 && pos < a.L_in && seg < a.B;
        hb[it] = 0;
        if (TWO_D) {
          int hh;
          const int item = fdiv(seg, a.h_n, inv_hn, hh);
          const int hrow0 = hh * a.h_stride - a.h_pad;
          hb[it] = ok ? hrow0 : -(1 << 20);
          vx1[it] = (unsigned)((item * cin * a.h_in + hrow0) * a.L_in + pos) * 4u;    // + kh rows, checked per load
          vx2[it] = RTG_OOB;
        } else {
          vx1[it] = ok ? (unsigned)(seg * a.C1 * a.L_in + pos) * 4u : RTG_OOB;
          vx2[it] = ok ? (unsigned)(seg * a.C2 * a.L_in + pos) * 4u : RTG_OOB;
        }
      }
      int dt;
      const int dseg = fdiv(v0 + lane, a.seg_len, a.inv_seg, dt);
      const bool colok = dseg < a.B && dt < a.Q;
      if (TWO_D) {
        int hh;
        const int item = fdiv(dseg, a.h_n, inv_hn, hh);
        vd = colok ? (unsigned)((item * rows_all * a.h_n + hh) * a.dy_L + dt) * 4u : RTG_OOB;
      } else {
        vd = colok ? (unsigned)(dseg * rows_all * a.dy_L + dt) * 4u : RTG_OOB;
      }
    } else {
      b0 = tl / a.n_ttiles;
      const int t0 = (tl - b0 * a.n_ttiles) * TT;
      const int o_start = t0 * a.stride - a.pad;
#pragma unroll
      for (int it = 0; it < MAXIT; ++it) {
        const int pos = o_start + lane + 64 * it;
        vx1[it] = (lane + 64 * it < a.PW && pos >= 0) ? (unsigned)pos * 4u : RTG_OOB;
        vx2[it] = vx1[it];
        hb[it] = 0;
      }
      vd = (unsigned)(t0 + lane) * 4u;
      item0 = b0;
      if (TWO_D) {
        item0 = b0 / a.h_n;
        hh0 = b0 - item0 * a.h_n;
      }
    }
    // ---- patch rows: consecutive channels, so the row base advances by one row pitch (2-D: by the rest of the
    // input plane when the kernel row wraps)
    if (TWO_D) {
      int kh = kh_first;
      const int hrow0 = hh0 * a.h_stride - a.h_pad;             // per-clip tiling: the input row of kernel row 0
      u64 q = (u64)a.x1 + (CONT ? (u64)ci_first * a.h_in
                                : (u64)(((long long)item0 * cin + ci_first) * a.h_in + hrow0 + kh)) * rowb;
      int left = a.x_bytes - ci_first * a.h_in * a.L_in * 4;    // continuous: bytes from the row base to the tensor end
      int rows_left = cw - wave * XR;                           // rows of the chunk still to stage (may be <= 0)
      chain32(rows_left); chain32(kh);
#pragma unroll
      for (int i = 0; i < XR; ++i) {
        const bool row_ok = rows_left > 0;
        --rows_left; chain32(rows_left);
        if (CONT) {
          const rsrc_t r = desc(q, row_ok ? left : 0);
          const unsigned khoff = (unsigned)(kh * a.L_in) * 4u;
#pragma unroll
          for (int it = 0; it < MAXIT; ++it) {
            const bool ok = (unsigned)(hb[it] + kh) < (unsigned)a.h_in;
            sx[i][it] = buf_load(r, ok ? vx1[it] + khoff : RTG_OOB);
          }
        } else {
          const int hrow = hrow0 + kh;
          const rsrc_t r = desc(q, (row_ok && hrow >= 0 && hrow < a.h_in) ? a.L_in * 4 : 0);
#pragma unroll
          for (int it = 0; it < MAXIT; ++it) sx[i][it] = buf_load(r, vx1[it]);
          q += rowb;
        }
        if (++kh == a.h_k) {
          kh = 0;
          q += (CONT ? (u64)a.h_in : (u64)(a.h_in - a.h_k)) * rowb;
          left -= a.h_in * a.L_in * 4;
        }
        chain(q); chain32(kh); chain32(left);
      }
    } else {
      const int gc0 = g * a.Cg + c_first;
      // two chains, one per source of a concatenated input; the second is only dereferenced for channels >= C1
      u64 q1 = (u64)a.x1 + (u64)((long long)(CONT ? 0 : b0) * a.C1 + gc0) * rowb;
      u64 q2 = (u64)a.x2 + (u64)((long long)(CONT ? 0 : b0) * a.C2 + gc0 - a.C1) * rowb;
      int left1 = a.x_bytes - gc0 * a.L_in * 4, left2 = x2_bytes - (gc0 - a.C1) * a.L_in * 4;
      int rows_left = cw - wave * XR;                           // rows of the chunk still to stage (may be <= 0)
      int first_left = a.C1 - gc0;                              // rows before the second source starts
      chain32(rows_left); chain32(first_left);
#pragma unroll
      for (int i = 0; i < XR; ++i) {
        const bool row_ok = rows_left > 0;
        const bool second = first_left <= 0;
        --rows_left; --first_left;
        chain32(rows_left); chain32(first_left);
        const int bytes = CONT ? (second ? left2 : left1) : a.L_in * 4;
        const rsrc_t r = desc(second ? q2 : q1, row_ok ? bytes : 0);
        if (second) {
#pragma unroll
          for (int it = 0; it < MAXIT; ++it) sx[i][it] = buf_load(r, vx2[it]);
        } else {
#pragma unroll
          for (int it = 0; it < MAXIT; ++it) sx[i][it] = buf_load(r, vx1[it]);
        }
        q1 += rowb; q2 += rowb;
        left1 -= a.L_in * 4; left2 -= a.L_in * 4;
        chain(q1); chain(q2); chain32(left1); chain32(left2);
      }
    }
    // ---- gy rows m_first, m_first + 4, ...: dy is [items, rows, h_n, dy_L] (h_n == 1 in 1-D)
    {
      const int gm = g * a.Mg + m_first;
      u64 d = (u64)a.dy + (CONT ? (u64)gm * dpitch
                                : (u64)((((long long)item0 * rows_all + gm) * a.h_n + hh0) * a.dy_L) * 4u);
      const u64 aux_delta = (u64)a.gy_aux - (u64)a.dy;
      int left = a.dy_bytes - gm * a.h_n * a.dy_L * 4;
      int m_left = a.Mg - m_first;                              // > 0 while the row exists
      chain32(m_left);
#pragma unroll
      for (int i = 0; i < DR; ++i) {
        const int bytes = (m_left > 0) ? (CONT ? left : a.Q * 4) : 0;
        m_left -= 4; chain32(m_left);
        sd[i] = buf_load(desc(d, bytes), vd);
        if (has_aux) sa[i] = buf_load(desc(d + aux_delta, bytes), vd);
        d += 4 * dpitch;
        left -= 4 * a.h_n * a.dy_L * 4;
        chain(d); chain32(left);
      }
    }
  };
  // branch-free leaky ReLU (slope 1 = identity): selects on loaded values tend to come back as exec-mask branches
  auto lrelu = [](float v, float slope) __attribute__((always_inline)) {
    return v > 0.f ? v : v * slope;
  };
  auto swrite = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < XR; ++i) {
      const int cl = wave * XR + i;
      if (__builtin_expect(cl < a.CKW, 1)) {
#pragma unroll
        for (int it = 0; it < MAXIT; ++it)
          if (lane + 64 * it < a.PW) {
            float v = sx[i][it];
            asm volatile("" : "+v"(v) : : "memory");   // consume the prefetched value here, below the MFMA loop
            xb[cl * a.ROW + lane + 64 * it] = lrelu(v, xslope);
          }
      }
    }
    float* dcol = db + wave * ROWD + lane;
    if (has_aux) {
      if (a.gy_mode == RTG_PRE_MUL_DTANH) {
#pragma unroll
        for (int i = 0; i < DR; ++i) {
          float v = sd[i], av = sa[i];
          asm volatile("" : "+v"(v), "+v"(av) : : "memory");
          dcol[4 * i * ROWD] = v * fmaf(-av, av, 1.f) * a.gy_scale;
        }
      } else {
#pragma unroll
        for (int i = 0; i < DR; ++i) {
          float v = sd[i], av = sa[i];
          asm volatile("" : "+v"(v), "+v"(av) : : "memory");
          // d/dx leaky_relu at av: 1 for av > 0, slope otherwise, as slope + (1 - slope) * [av > 0]
          dcol[4 * i * ROWD] = v * fmaf(1.f - a.gy_slope, (float)(av > 0.f), a.gy_slope) * a.gy_scale;
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < DR; ++i) {
        float v = sd[i];
        asm volatile("" : "+v"(v) : : "memory");
        dcol[4 * i * ROWD] = lrelu(v, gslope) * a.gy_scale;
      }
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
    if constexpr (BF) {
      constexpr int KS = KK * 4;                 // reduction positions per bf16 MFMA: 8 (32x32x8) or 16 (16x16x16)
      constexpr int GB = 4;                      // MFMA k-steps per read phase
#pragma unroll
      for (int t0 = 0; t0 < TT; t0 += GB * KS) {
        bf4 af[MTW][GB], bfv[NTW][GB];
#pragma unroll
        for (int u = 0; u < GB; ++u) {
          const int t = t0 + u * KS;
#pragma unroll
          for (int i = 0; i < MTW; ++i) af[i][u] = to_bf4(ap[i][t], ap[i][t + 1], ap[i][t + 2], ap[i][t + 3]);
#pragma unroll
          for (int j = 0; j < NTW; ++j)
            bfv[j][u] = to_bf4(bp[j][t * a.stride], bp[j][(t + 1) * a.stride], bp[j][(t + 2) * a.stride],
                               bp[j][(t + 3) * a.stride]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < GB; ++u)
#pragma unroll
          for (int i = 0; i < MTW; ++i)
#pragma unroll
            for (int j = 0; j < NTW; ++j) acc[i][j] = M::runbf(af[i][u], bfv[j][u], acc[i][j]);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else
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

template <int TM, int MTW, int NTW, int WM, int MAXIT, int MODE>
__global__ __launch_bounds__(RTG_THREADS) void wgrad_kernel(const WgArgs a) {
  wgrad_body<TM, MTW, NTW, WM, MAXIT, MODE>(a, blockIdx.x);
}

// Several weight-gradient problems of ONE kernel instance in one launch (the parallel ResBlock branches of a UNet-G stage,
// the six convs of a ResidualStack): a block finds its member by its index; every member's grid is a multiple of 8 blocks,
// so the XCD a block runs on is the one its member's own launch would have given it.
struct WgGroupArgs {
  int n;
  unsigned blk_end[RTG_WG_MAX_GROUP];
  WgArgs p[RTG_WG_MAX_GROUP];
};

template <int TM, int MTW, int NTW, int WM, int MAXIT, int MODE>
__global__ __launch_bounds__(RTG_THREADS) void wgrad_group_kernel(const WgGroupArgs ga) {
  int pid = 0;
  unsigned start = 0;
  for (int i = 0; i + 1 < ga.n; ++i)
    if (blockIdx.x >= ga.blk_end[i]) {
      pid = i + 1;
      start = ga.blk_end[i];
    }
  wgrad_body<TM, MTW, NTW, WM, MAXIT, MODE>(ga.p[pid], blockIdx.x - start);
}

template <int TM, int MTW, int NTW, int WM, int MODE>
int launch_group(const WgGroupArgs& ga, size_t lds_bytes, hipStream_t s) {
  auto k = wgrad_group_kernel<TM, MTW, NTW, WM, 2, MODE>;          // staging width of the members: PW <= 128 (host check)
  static std::atomic<unsigned> optin{0};              // (> 64 KB of dynamic LDS: opt-in per kernel and device)
  if (lds_bytes > 64 * 1024 && rtg_lds_optin((const void*)k, optin) != RTG_OK) return RTG_ERANGE;
  RTG_KLAUNCH(k, dim3(ga.blk_end[ga.n - 1]), dim3(RTG_THREADS), lds_bytes, s, ga);
  return rtg_launch_status();
}

template <int MODE>
int launch_group_mode(int shape, const WgGroupArgs& ga, size_t lds_bytes, hipStream_t s) {
  switch (shape) {                                                   // 32-row MFMA tiles only
    case 0: return launch_group<32, 2, 2, 2, MODE>(ga, lds_bytes, s);
    case 1: return launch_group<32, 2, 2, 1, MODE>(ga, lds_bytes, s);
    case 2: return launch_group<32, 1, 2, 1, MODE>(ga, lds_bytes, s);
    case 3: return launch_group<32, 1, 4, 1, MODE>(ga, lds_bytes, s);
    case 4: return launch_group<32, 1, 1, 1, MODE>(ga, lds_bytes, s);
    default: return launch_group<32, 1, 1, 4, MODE>(ga, lds_bytes, s);
  }
}

template <int TM, int MTW, int NTW, int WM, int MAXIT, int MODE>
int launch(const WgArgs& a, dim3 grid, size_t lds_bytes, hipStream_t s) {
  auto k = wgrad_kernel<TM, MTW, NTW, WM, MAXIT, MODE>;
  static std::atomic<unsigned> optin{0};              // (> 64 KB of dynamic LDS: opt-in per kernel and device)
  if (lds_bytes > 64 * 1024 && rtg_lds_optin((const void*)k, optin) != RTG_OK) return RTG_ERANGE;
  RTG_KLAUNCH(k, grid, dim3(RTG_THREADS), lds_bytes, s, a);
  return rtg_launch_status();
}

template <int TM, int MTW, int NTW, int WM, int MODE>
int launch_it(int maxit, const WgArgs& a, dim3 grid, size_t lds_bytes, hipStream_t s) {
  if (maxit <= 2) return launch<TM, MTW, NTW, WM, 2, MODE>(a, grid, lds_bytes, s);
  if (maxit <= 4) return launch<TM, MTW, NTW, WM, 4, MODE>(a, grid, lds_bytes, s);
  return launch<TM, MTW, NTW, WM, RTG_PW_MAX / 64, MODE>(a, grid, lds_bytes, s);
}

template <int TM, int MODE>
int launch_shape(int shape, int maxit, const WgArgs& a, dim3 grid, size_t lds_bytes, hipStream_t s) {
  switch (shape) {
    case 0: return launch_it<TM, 2, 2, 2, MODE>(maxit, a, grid, lds_bytes, s);
    case 1: return launch_it<TM, 2, 2, 1, MODE>(maxit, a, grid, lds_bytes, s);
    case 2: return launch_it<TM, 1, 2, 1, MODE>(maxit, a, grid, lds_bytes, s);
    case 3: return launch_it<TM, 1, 4, 1, MODE>(maxit, a, grid, lds_bytes, s);
    case 4: return launch_it<TM, 1, 1, 1, MODE>(maxit, a, grid, lds_bytes, s);
    default: return launch_it<TM, 1, 1, 4, MODE>(maxit, a, grid, lds_bytes, s);
  }
}


// entry of one addressing mode: dispatch on (MFMA tile, block shape, staging width)
template <int MODE>
int launch_mode(int tm, int shape, int maxit, const WgArgs& a, dim3 grid, size_t lds_bytes, hipStream_t s) {
  if (tm == 32) return launch_shape<32, MODE>(shape, maxit, a, grid, lds_bytes, s);
  return launch_shape<16, MODE>(shape, maxit, a, grid, lds_bytes, s);
}

}  // namespace rtg_wg

#define RTG_WGRAD_DEFINE_GROUP(N)                                                                                  \
  int rtg_wgrad_launch_group_m##N(int shape, const rtg_wg::WgGroupArgs& ga, size_t lds_bytes, hipStream_t s) {      \
    return rtg_wg::launch_group_mode<N>(shape, ga, lds_bytes, s);                                                    \
  }

#define RTG_WGRAD_DEFINE_MODE(N)                                                                                   \
  int rtg_wgrad_launch_m##N(int tm, int shape, int maxit, const rtg_wg::WgArgs& a, dim3 grid, size_t lds_bytes,     \
                            hipStream_t s) {                                                                         \
    return rtg_wg::launch_mode<N>(tm, shape, maxit, a, grid, lds_bytes, s);                                          \
  }
