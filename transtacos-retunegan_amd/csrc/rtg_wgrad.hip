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
// A block owns (group, MB m-tiles, one channel chunk, one split of the reduction) and walks its share of the tile list
// with register-prefetched double buffering (buffer loads with hardware bounds checking: branch-free, all in flight
// under the MFMA loop); its 4 waves share the staged tiles and each owns up to TPW accumulator tiles.  Partials are
// stored per split (fixed-order reduction in rtg_weightnorm_backward).
#include "rtg_common.h"

namespace {

constexpr int TT = 64;          // reduction (virtual position) steps per staged tile
constexpr int ROWD = 81;        // LDS pitch of the gy tile: odd, and 81^-1 = 17 (mod 32) keeps 16-row reads conflict free
constexpr int TPW = 4;          // accumulator tiles per wave
constexpr int MAXROWS = 64;     // gy rows staged per block

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
  int CKW, n_cchunk, MB, m_blocks, NTB, n_ttiles, n_tiles_total, PW, ROW, ones_off;
  int seg_len, seg_pitch, seg_nb, seg_pw;   // seg_len == 0: one clip per tile
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

template <int TM, int MAXIT>   // MAXIT: 64-float pieces of a patch row each lane stages (>= PW / 64)
__global__ __launch_bounds__(RTG_THREADS) void wgrad_kernel(const WgArgs a) {
  using M = MfmaW<TM>;
  using acc_t = typename M::acc_t;
  constexpr int KK = 64 / TM;
  constexpr int XR = RTG_CK / 4;         // patch rows per wave (CKW <= 16)
  constexpr int DR = MAXROWS / 4;        // gy rows per wave

  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // block coordinates
  int by = blockIdx.y;
  const int cchunk = by % a.n_cchunk; by /= a.n_cchunk;
  const int mb = by % a.m_blocks;
  const int g = by / a.m_blocks;
  const int split = blockIdx.x;
  const int c0 = cchunk * a.CKW;
  const int cw = min(a.CKW, a.Cg - c0);
  const int rows_blk = a.MB * TM;
  const int m0 = mb * rows_blk;                      // first row (within group) of this block
  const bool packed = a.seg_len > 0;

  const int xbuf_sz = a.CKW * a.ROW;                 // floats
  const int dbuf_sz = rows_blk * ROWD;
  float* ones = lds + a.ones_off;
  for (int i = tid; i < TT * a.stride + 8; i += RTG_THREADS) ones[i] = 1.f;

  // per-wave accumulator tiles
  const int n_tiles = a.MB * a.NTB;
  acc_t acc[TPW];
  int a_base[TPW], b_base[TPW];
  const int n_lane = lane & (TM - 1), kk = lane / TM;
#pragma unroll
  for (int k = 0; k < TPW; ++k) {
#pragma unroll
    for (int r = 0; r < M::NREG; ++r) acc[k][r] = 0.f;
    const int tile = wave + 4 * k;
    const int mt = tile / a.NTB, nt = tile - mt * a.NTB;
    a_base[k] = (mt * TM + n_lane) * ROWD + kk;
    const int n = nt * TM + n_lane;                  // column within the chunk
    const int cl = n / a.K, j = n - cl * a.K;
    b_base[k] = (cl < a.CKW) ? (cl * a.ROW + j * a.dil + kk * a.stride) : (-(1 << 20) + kk * a.stride);
  }

  // ---- static staging geometry
  // x patch element o = lane + 64*it of a row: (segment, position within the clip's patch)
  int xseg[MAXIT], xw[MAXIT];
#pragma unroll
  for (int it = 0; it < MAXIT; ++it) {
    const int o = lane + 64 * it;
    if (packed) {
      xseg[it] = o / a.seg_pitch;
      xw[it] = o - xseg[it] * a.seg_pitch;
      if (xseg[it] >= a.seg_nb || xw[it] >= a.seg_pw || o >= a.PW) xw[it] = -(1 << 28);   // never valid
    } else {
      xseg[it] = 0;
      xw[it] = (o < a.PW) ? o : -(1 << 28);
    }
  }
  // gy tile column = lane: (segment, position within the clip)
  const int dseg = packed ? lane / a.seg_len : 0;
  const int dt = packed ? lane - dseg * a.seg_len : lane;
  const bool dcol_ok = packed ? (dseg < a.seg_nb) : true;

  const rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc((void*)a.x1, 0, a.B * a.C1 * a.L_in * 4, 0x00020000);
  const rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x2 ? a.x2 : a.x1), 0,
                                                      a.x2 ? a.B * a.C2 * a.L_in * 4 : 0, 0x00020000);
  const int dy_bytes = a.B * a.groups * a.Mg * a.dy_L * 4;
  const rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, dy_bytes, 0x00020000);
  const rsrc_t raux = __builtin_amdgcn_make_buffer_rsrc((void*)(a.gy_aux ? a.gy_aux : a.dy), 0,
                                                        a.gy_aux ? dy_bytes : 0, 0x00020000);
  const bool has_aux = a.gy_aux != nullptr && (a.gy_mode == RTG_PRE_MUL_DLRELU || a.gy_mode == RTG_PRE_MUL_DTANH);
  const float xslope = (a.pre_mode == RTG_PRE_LRELU) ? a.pre_slope : 1.f;
  const float gslope = (a.gy_mode == RTG_PRE_LRELU) ? a.gy_slope : 1.f;

  float sx[XR][MAXIT], sd[DR], sa[DR];

  auto gload = [&](int tl) __attribute__((always_inline)) {
    int b0, t0;
    if (packed) { b0 = tl * a.seg_nb; t0 = 0; }
    else        { b0 = tl / a.n_ttiles; t0 = (tl - b0 * a.n_ttiles) * TT; }
    const int o_start = t0 * a.stride - a.pad;
    // input patch: rows c0 + wave*XR + i
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
    // gy tile: rows m0 + wave + 4*i, column = lane
    {
      const int bb = b0 + dseg;
      const int t = t0 + dt;
      const bool colok = dcol_ok && bb < a.B && t < a.Q;
      const unsigned coloff = colok ? ((unsigned)bb * (unsigned)(a.groups * a.Mg) * (unsigned)a.dy_L + (unsigned)t) * 4u
                                    : RTG_OOB;
#pragma unroll
      for (int i = 0; i < DR; ++i) {
        const int rl = wave + 4 * i;
        const int m = m0 + rl;
        const bool rok = rl < rows_blk && m < a.Mg;
        const unsigned off = rok ? (coloff + (unsigned)(g * a.Mg + m) * (unsigned)a.dy_L * 4u) | (coloff & RTG_OOB)
                                 : RTG_OOB;
        sd[i] = buf_load(rdy, off);
        if (has_aux) sa[i] = buf_load(raux, off);
      }
    }
  };
  auto swrite = [&](int which) __attribute__((always_inline)) {
    float* xb = lds + which * (xbuf_sz + dbuf_sz);
    float* db = xb + xbuf_sz;
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
      if (rl < rows_blk) {
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
    }
  };

  const int total = a.n_tiles_total;
  int tl = split;
  int which = 0;
  if (tl < total) {
    gload(tl);
    swrite(0);
  }
  __syncthreads();
  for (; tl < total; tl += a.splits) {
    const float* xb = lds + which * (xbuf_sz + dbuf_sz);
    const float* db = xb + xbuf_sz;
    const bool more = tl + a.splits < total;
    if (more) gload(tl + a.splits);
#pragma unroll
    for (int k = 0; k < TPW; ++k) {
      if (wave + 4 * k >= n_tiles) continue;
      const float* ap = db + a_base[k];
      const float* bp = (b_base[k] >= 0) ? xb + b_base[k] : ones + (b_base[k] + (1 << 20));
      acc_t c = acc[k];
      // read phase / MFMA phase in groups of 16 k-steps (distinct registers, so the reads are all in flight before
      // the first MFMA; the other wave on the SIMD computes meanwhile)
      constexpr int GRP = 16;
#pragma unroll
      for (int t0 = 0; t0 < TT; t0 += GRP * KK) {
        float af[GRP], bf[GRP];
#pragma unroll
        for (int u = 0; u < GRP; ++u) {
          af[u] = ap[t0 + u * KK];
          bf[u] = bp[(t0 + u * KK) * a.stride];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < GRP; ++u) c = M::run(af[u], bf[u], c);
        __builtin_amdgcn_sched_barrier(0);
      }
      acc[k] = c;
    }
    if (more) swrite(which ^ 1);
    __syncthreads();
    which ^= 1;
  }

  // store this split's partial
  float* wpart = a.part + (size_t)split * a.part_stride;
  float* bpart = wpart + (size_t)a.groups * a.Mg * a.Cg * a.K;
  const int nb = a.CKW * a.K;                        // the "ones" column
#pragma unroll
  for (int k = 0; k < TPW; ++k) {
    const int tile = wave + 4 * k;
    if (tile >= n_tiles) continue;
    const int mt = tile / a.NTB, nt = tile - mt * a.NTB;
    const int n = nt * TM + n_lane;
#pragma unroll
    for (int r = 0; r < M::NREG; ++r) {
      const int m = m0 + mt * TM + M::row(lane, r);
      if (m >= a.Mg) continue;
      const size_t rowg = (size_t)g * a.Mg + m;
      if (n < cw * a.K) wpart[rowg * (a.Cg * a.K) + (size_t)c0 * a.K + n] = acc[k][r];
      else if (n == nb && cchunk == 0) bpart[rowg] = acc[k][r];
    }
  }
}

struct WgGeom {
  int TM, CKW, MB, NTB, n_cchunk, m_blocks, n_ttiles, n_tiles_total, PW, ROW;
  int seg_len, seg_nb, seg_pw;
};

int geometry(const RtgWgradDesc* d, WgGeom* o) {
  const int TM = d->Mg >= 32 ? 32 : 16;
  const int n_mt = rtg_ceil_div(d->Mg, TM);
  int CKW = d->Cg < RTG_CK ? d->Cg : RTG_CK;
  // shrink the channel chunk until the chunk's columns (+1 bias column) fit 4*TPW tiles with at least one m-tile
  while (CKW > 1 && rtg_ceil_div(CKW * d->K + 1, TM) > 4 * TPW) CKW = (CKW + 1) / 2;
  const int NTB = rtg_ceil_div(CKW * d->K + 1, TM);
  if (NTB > 4 * TPW) return RTG_ERANGE;
  int MB = (4 * TPW) / NTB;
  if (MB > n_mt) MB = n_mt;
  if (MB * TM > MAXROWS) MB = MAXROWS / TM;
  if (MB < 1) MB = 1;
  o->TM = TM; o->CKW = CKW; o->MB = MB; o->NTB = NTB;
  o->n_cchunk = rtg_ceil_div(d->Cg, CKW);
  o->m_blocks = rtg_ceil_div(n_mt, MB);
  o->PW = (TT - 1) * d->stride + (d->K - 1) * d->dil + 1;
  if (o->PW > RTG_PW_MAX) return RTG_ERANGE;
  // segment packing for short rows
  const int extra = (d->K - 1) * d->dil + 1 - d->stride;
  const int Lseg = d->Q + (extra > 0 ? (extra + d->stride - 1) / d->stride : 0);
  o->seg_pw = (d->Q - 1) * d->stride + (d->K - 1) * d->dil + 1;
  if (2 * Lseg <= TT && d->B >= 2) {
    o->seg_len = Lseg;
    o->seg_nb = TT / Lseg < d->B ? TT / Lseg : d->B;
    o->n_ttiles = 1;
    o->n_tiles_total = rtg_ceil_div(d->B, o->seg_nb);
  } else {
    o->seg_len = 0;
    o->seg_nb = 1;
    o->n_ttiles = rtg_ceil_div(d->Q, TT);
    o->n_tiles_total = d->B * o->n_ttiles;
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
  // 32-bit buffer offsets
  if ((long long)d->B * (d->C1 > d->C2 ? d->C1 : d->C2) * d->L_in * 4 >= (1ll << 31)) return RTG_ERANGE;
  if ((long long)d->B * d->groups * d->Mg * d->dy_L * 4 >= (1ll << 31)) return RTG_ERANGE;
  return RTG_OK;
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
  long long s = (768 + base - 1) / base;         // aim at ~3 blocks per CU
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

  WgArgs a;
  a.x1 = x1; a.x2 = x2; a.dy = dy; a.gy_aux = gy_aux; a.part = part;
  a.B = d->B; a.C1 = d->C1; a.C2 = d->C2; a.L_in = d->L_in; a.groups = d->groups; a.Cg = d->Cg; a.Mg = d->Mg;
  a.K = d->K; a.stride = d->stride; a.dil = d->dil; a.pad = d->pad; a.Q = d->Q; a.dy_L = d->dy_L;
  a.pre_mode = d->pre_mode; a.pre_slope = d->pre_slope; a.gy_mode = d->gy_mode; a.gy_slope = d->gy_slope;
  a.gy_scale = d->gy_scale;
  a.splits = d->splits; a.part_stride = d->part_stride;
  a.CKW = g.CKW; a.n_cchunk = g.n_cchunk; a.MB = g.MB; a.m_blocks = g.m_blocks; a.NTB = g.NTB;
  a.n_ttiles = g.n_ttiles; a.n_tiles_total = g.n_tiles_total; a.PW = g.PW; a.ROW = g.ROW;
  a.seg_len = g.seg_len; a.seg_nb = g.seg_nb; a.seg_pw = g.seg_pw;
  a.seg_pitch = g.seg_len > 0 ? g.seg_len * d->stride : 1;
  const int buf = g.CKW * g.ROW + g.MB * g.TM * ROWD;
  a.ones_off = 2 * buf;
  const size_t lds_bytes = (size_t)(2 * buf + TT * d->stride + 8) * sizeof(float);
  const long long gy = (long long)d->groups * g.m_blocks * g.n_cchunk;
  if (gy > 65535) return RTG_ERANGE;
  dim3 grid(d->splits, (unsigned)gy, 1);
  hipStream_t s = (hipStream_t)stream;
#define RTG_WG(tm, mi)                                                                                           \
  {                                                                                                              \
    auto k = wgrad_kernel<tm, mi>;                                                                               \
    if (lds_bytes > 64 * 1024)                                                                                   \
      hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);           \
    hipLaunchKernelGGL(k, grid, dim3(RTG_THREADS), lds_bytes, s, a);                                             \
    return rtg_launch_status();                                                                                  \
  }
  if (g.TM == 32) {
    if (g.PW <= 2 * 64) RTG_WG(32, 2)
    if (g.PW <= 4 * 64) RTG_WG(32, 4)
    RTG_WG(32, RTG_PW_MAX / 64)
  } else {
    if (g.PW <= 2 * 64) RTG_WG(16, 2)
    if (g.PW <= 4 * 64) RTG_WG(16, 4)
    RTG_WG(16, RTG_PW_MAX / 64)
  }
#undef RTG_WG
}
