// rtg_resconv.hip — weights-in-registers MFMA kernel for the stride-1 "same" convolutions of the UNet-G residual blocks.
//
// Serves (forward and backward-data alike, through the same packed weights rtg_conv1d uses):
//   ResBlock3      x <- conv_d(lrelu(x)) + x,  C in {32, 64}, k in {3, 5, 7}, d in {9, 3, 1}   generator.py:133-155
//   ResidualStack  r = conv_d(lrelu(x)); x <- conv_1(lrelu(r)) + x, k = 3                       generator.py:33-77
// i.e. groups == 1, C_in == C_out in {32, 64}, stride 1, L_out == L_in, L % 4 == 0.  These layers are 70 % of the UNet-G
// MACs but short in K (96 .. 448): the general kernel (rtg_conv1d_kernel.h) spends 35-45 % of a workgroup's life on its
// fixed costs there.  Here
//   * a wave keeps ALL A fragments of its 32 output rows in registers (C/2 * k <= 224 VGPRs, loaded once per block),
//   * a block walks several position tiles; the raw input window of tile i+1 (C rows x (tile + halo) samples) travels
//     global -> registers -> LDS while tile i is multiplied (double-buffered LDS, one barrier per tile),
//   * the input activation is applied when a B fragment is read (1 ds_read_b32 + 3 VALU per 64-cycle MFMA), so the LDS
//     holds the raw samples and the residual x of ResBlock3 comes from LDS instead of a second global read,
//   * the inner loop is 1 LDS read + 1 MFMA: v_mfma_f32_32x32x2_f32 issues back to back.
// The accumulation order (16-channel chunk, tap, channel pair) and the epilogue arithmetic are those of the general
// kernel: results are bit-identical to every other block shape (tests/test_conv1d_gpu.py), so the tuner may pick freely.
// Exposed through rtg_conv1d as block-shape codes 7001 / 7002 (RtgConv1dDesc.tile_cfg): 32 / 64 positions per wave.
#include "rtg_common.h"
#include <stdlib.h>

namespace {

struct ResArgs {
  const float *x, *wp, *bias, *mask, *res;
  float* out;
  int B, C, L, dil, pad, m, Wl, Wp, n_t, total;
  int pre, res_in, act, dbg;
  float pre_slope, mask_slope, out_scale, act_slope;
};

__device__ __forceinline__ int mrow32(int lane, int r) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// blocks per CU the registers allow: two where the A fragments + one column tile fit in 256 registers per lane (then one
// block's staging / epilogue overlaps the other's multiplications), else one
template <int CIN, int KT, int NT>
constexpr int kResconvOcc = (NT == 1 && CIN == 32) ? 2 : 1;

template <int CIN, int KT, int NT>
__global__ __launch_bounds__(RTG_THREADS, (kResconvOcc<CIN, KT, NT>)) void resconv_kernel(const ResArgs a) {
  constexpr int NCC = CIN / RTG_CK, NA = NCC * KT * 8;
  constexpr int PT = 128 * NT, TPROW = NT == 2 ? 128 : 64, RP = RTG_THREADS / TPROW, NLD = CIN / RP;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kk = lane >> 5, n_lane = lane & 31;
  const int mt = blockIdx.y;

  const int tpb = (a.total + (int)gridDim.x - 1) / (int)gridDim.x;
  const int tile_lo = blockIdx.x * tpb;
  const int tile_hi = min(a.total, tile_lo + tpb);
  if (tile_lo >= tile_hi) return;

  // ---- staging: thread = float4 column q of rows rr, rr + RP, ... (aligned 16-byte loads and LDS writes: LDS column 0
  // is input position a0 = floor4(t0 - pad); positions outside the clip are zeros)
  const int q = tid & (TPROW - 1), rr = tid / TPROW;
  const bool qok = 4 * q < a.Wl;
  f32x4 pf[NLD];
  auto issue = [&](int tile) __attribute__((always_inline)) {
    const int b = tile / a.n_t, t0 = (tile - b * a.n_t) * PT;
    const int pos = ((t0 - a.pad) & ~3) + 4 * q;
    const bool ok = qok && pos >= 0 && pos < a.L && !(a.dbg & 4);
    const float* src = a.x + ((size_t)b * CIN + rr) * a.L + pos;
#pragma unroll
    for (int u = 0; u < NLD; ++u)
      pf[u] = ok ? *reinterpret_cast<const f32x4*>(src + (size_t)u * RP * a.L) : f32x4{0.f, 0.f, 0.f, 0.f};
  };
  auto commit = [&](float* buf) __attribute__((always_inline)) {
    if (qok) {
#pragma unroll
      for (int u = 0; u < NLD; ++u) *reinterpret_cast<f32x4*>(buf + (rr + u * RP) * a.Wp + 4 * q) = pf[u];
    }
  };
  issue(tile_lo);

  // ---- A fragments of this block's 32 output rows: ((chunk, tap), channel pair) -> 64 consecutive floats each
  float A[NA];
  {
    const float* wpm = a.wp + (size_t)mt * NA * 64 + lane;
#pragma unroll
    for (int f = 0; f < NA; ++f) A[f] = wpm[f * 64];
  }
  const int bufsz = CIN * a.Wp;
  commit(lds);
  __syncthreads();

  const int colbase = kk * a.Wp + a.m + wave * (32 * NT) + n_lane;     // B-fragment column of tap 0, tile nt = 0
  const float mslope = a.mask ? a.mask_slope : 1.f;
  const float pslope = a.pre ? a.pre_slope : 1.f;                      // (x * 1.0f == x: no branch in the loop)
  // epilogue geometry: the wave's [32 rows][32 * NT positions] tile is transposed through LDS so that a lane owns four
  // consecutive positions of one row: 16-byte loads / stores, a whole row of the tile per 8 * NT lanes
  constexpr int C4 = 8 * NT;                      // float4 columns per row of the wave tile
  constexpr int RPI = 64 / C4;                    // rows per pass
  constexpr int NPASS = 32 / RPI;
  float* scr = lds + 2 * bufsz + wave * (32 * 32 * NT);
  const int erow = lane / C4, ecol = lane % C4;
  const int Cout = gridDim.y * 32;
  for (int tile = tile_lo, it = 0; tile < tile_hi; ++tile, ++it) {
    const bool next = tile + 1 < tile_hi;
    if (next) issue(tile + 1);
    const float* buf = lds + (it & 1) * bufsz;
    const int b = tile / a.n_t, t0 = (tile - b * a.n_t) * PT;
    const int pl0 = wave * (32 * NT) + 4 * ecol;                       // first of the lane's four positions in the tile
    const int pos0 = t0 + pl0;
    const bool pok = pos0 < a.L;                                       // (L % 4 == 0: the four are in or out together)
    // epilogue operand of this tile (leaky-relu-derivative mask, or a residual that is not the conv input) requested
    // before the multiplications: its latency hides behind them
    f32x4 ep[NPASS];
    const float* eptr = a.mask ? a.mask : ((a.res && !a.res_in) ? a.res : nullptr);
    if (eptr) {
#pragma unroll
      for (int i = 0; i < NPASS; ++i) {
        const size_t o = ((size_t)b * Cout + mt * 32 + i * RPI + erow) * a.L + pos0;
        ep[i] = pok ? *reinterpret_cast<const f32x4*>(eptr + o) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    f32x16 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    // ---- multiply: one LDS read + one MFMA per (chunk, tap, channel pair, column tile); the B values of step s + 2 are
    // requested before the MFMAs of step s
    {
      const float* bp = buf + colbase;
      constexpr int NS = NCC * KT * 8;
      constexpr int PD = 2;
      float vb[PD + 1][NT];
      auto bload = [&](int s_, float (&dst)[NT]) __attribute__((always_inline)) {
        const int cc = s_ / (KT * 8), tap = (s_ / 8) % KT, cp = s_ % 8;
        const float* brow = bp + tap * a.dil + (cc * RTG_CK + cp * 2) * a.Wp;
#pragma unroll
        for (int j = 0; j < NT; ++j) dst[j] = brow[j * 32];
      };
#pragma unroll
      for (int s_ = 0; s_ < PD; ++s_) bload(s_, vb[s_]);
#pragma unroll
      for (int s_ = 0; s_ < NS; ++s_) {
        if (s_ + PD < NS) bload(s_ + PD, vb[(s_ + PD) % (PD + 1)]);
        __builtin_amdgcn_sched_barrier(0);            // (keeps the request ahead of the MFMAs: the compiler sinks it otherwise)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          float v = vb[s_ % (PD + 1)][j];
          v = v > 0.f ? v : v * pslope;
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[s_], v, acc[j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // ---- epilogue: bias, mask, residual, scale, activation (the order and arithmetic of rtg_conv1d_kernel.h)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) scr[mrow32(lane, r) * (32 * NT) + j * 32 + n_lane] = acc[j][r];
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
      const int row = i * RPI + erow, mrow = mt * 32 + row;
      const f32x4 t = *reinterpret_cast<const f32x4*>(scr + row * (32 * NT) + 4 * ecol);
      float v[4] = {t.x, t.y, t.z, t.w};
      const float bias_v = a.bias ? a.bias[mrow] : 0.f;
      float mk[4] = {1.f, 1.f, 1.f, 1.f}, rv[4] = {0.f, 0.f, 0.f, 0.f};
      if (a.mask) { mk[0] = ep[i].x; mk[1] = ep[i].y; mk[2] = ep[i].z; mk[3] = ep[i].w; }
      if (a.res_in) {
        const f32x4 r4 = *reinterpret_cast<const f32x4*>(buf + mrow * a.Wp + a.m + a.pad + pl0);   // (m + pad) % 4 == 0
        rv[0] = r4.x; rv[1] = r4.y; rv[2] = r4.z; rv[3] = r4.w;
      } else if (a.res) {
        f32x4 r4 = ep[i];
        if (a.mask) r4 = pok ? *reinterpret_cast<const f32x4*>(a.res + ((size_t)b * Cout + mrow) * a.L + pos0) : f32x4{0.f, 0.f, 0.f, 0.f};
        rv[0] = r4.x; rv[1] = r4.y; rv[2] = r4.z; rv[3] = r4.w;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float w_ = v[e] + bias_v;
        w_ = __builtin_fmaf(w_, mk[e] > 0.f ? 1.f : mslope, rv[e]) * a.out_scale;
        if (a.act == RTG_ACT_LRELU) w_ = rtg_lrelu(w_, a.act_slope);
        v[e] = w_;
      }
      if (pok && !(a.dbg & 2))
        *reinterpret_cast<f32x4*>(a.out + ((size_t)b * Cout + mrow) * a.L + pos0) = f32x4{v[0], v[1], v[2], v[3]};
    }
    if (next) commit(lds + ((it + 1) & 1) * bufsz);
    __syncthreads();
  }
}

template <int CIN, int KT, int NT>
int launch(ResArgs a, int n_mt, size_t lds_bytes, hipStream_t s) {
  // as many blocks as fill the chip at the kernel's occupancy, each a run of tiles
  int gx = 256 * kResconvOcc<CIN, KT, NT> / n_mt;
  if (gx > a.total) gx = a.total;
  gx = rtg_ceil_div(a.total, rtg_ceil_div(a.total, gx));
  static std::atomic<unsigned> optin{0};              // (> 64 KB of dynamic LDS: opt-in per kernel and device)
  if (rtg_lds_optin(reinterpret_cast<const void*>(&resconv_kernel<CIN, KT, NT>), optin) != RTG_OK) return RTG_ERANGE;
  RTG_KLAUNCH((resconv_kernel<CIN, KT, NT>), dim3((unsigned)gx, n_mt), dim3(RTG_THREADS), lds_bytes, s, a);
  return rtg_launch_status();
}

}  // namespace

// block-shape codes of this kernel in RtgConv1dDesc.tile_cfg
#define RTG_RESCONV_CODE 7000

// 0: the descriptor is not a shape this kernel serves; else a bit mask of the NT variants (bit 0: NT = 1, bit 1: NT = 2)
int rtg_resconv_variants(const RtgConv1dDesc* d) {
  if (d->groups != 1 || d->C2 != 0 || d->stride != 1 || d->shuf_S != 1 || d->out_split != 0 || d->accumulate) return 0;
  if (d->h_k > 1 || d->h_n > 1 || d->tap_major || d->bf16 || d->tile_m != 32) return 0;
  if (d->Cg != d->Mg || (d->Cg != 32 && d->Cg != 64) || d->C1 != d->Cg || d->out_C != d->Mg) return 0;
  if (d->K != 3 && d->K != 5 && d->K != 7) return 0;
  if (d->Q != d->L_in || d->out_L != d->L_in || d->L_in % 4 != 0 || d->L_in < 32) return 0;
  if (d->pad < 0 || d->pad > (d->K - 1) * d->dil || d->dil > 16) return 0;
  if (d->pre_mode != RTG_PRE_NONE && d->pre_mode != RTG_PRE_LRELU) return 0;
  if (d->act != RTG_ACT_NONE && d->act != RTG_ACT_LRELU) return 0;
  return d->Cg == 32 ? 3 : 1;
}

int rtg_resconv_launch(const RtgConv1dDesc* d, int nt, const float* x, const float* wp, const float* bias,
                       const float* mask, const float* res, float* out, hipStream_t s) {
  const int var = rtg_resconv_variants(d);
  if (!(var & (1 << (nt - 1)))) return RTG_EINVAL;
  if (!x || !wp || !out) return RTG_ENULL;
  if ((reinterpret_cast<uintptr_t>(x) & 15) != 0) return RTG_EINVAL;
  ResArgs a;
  a.x = x; a.wp = wp; a.bias = bias; a.mask = mask; a.res = res; a.out = out;
  a.B = d->B; a.C = d->Cg; a.L = d->L_in; a.dil = d->dil; a.pad = d->pad;
  const int PT = 128 * nt;
  a.m = (-d->pad) & 3;                                 // (t0 - pad) mod 4, t0 a multiple of 128
  a.Wl = (a.m + PT + (d->K - 1) * d->dil + 3) & ~3;
  a.Wp = a.Wl + 4;
  if (a.Wl > 4 * (nt == 2 ? 128 : 64)) return RTG_ERANGE;
  a.n_t = rtg_ceil_div(d->L_in, PT);
  a.total = d->B * a.n_t;
  a.pre = d->pre_mode == RTG_PRE_LRELU ? 1 : 0;
  a.pre_slope = d->pre_slope;
  a.res_in = (res != nullptr && res == x) ? 1 : 0;
  a.act = d->act; a.act_slope = d->act_slope;
  a.mask_slope = d->mask_slope; a.out_scale = d->out_scale;
  const int n_mt = d->Mg / 32;
  const size_t lds_bytes = ((size_t)2 * d->Cg * a.Wp + 4 * 32 * 32 * nt) * sizeof(float);   // 2 windows + transposes
  if (lds_bytes > 160 * 1024) return RTG_ERANGE;
  a.dbg = 0;
#define RTG_RC(c, k, n) \
  if (d->Cg == c && d->K == k && nt == n) return launch<c, k, n>(a, n_mt, lds_bytes, s);
  RTG_RC(32, 3, 1) RTG_RC(32, 5, 1) RTG_RC(32, 7, 1) RTG_RC(32, 3, 2) RTG_RC(32, 5, 2) RTG_RC(32, 7, 2)
  RTG_RC(64, 3, 1) RTG_RC(64, 5, 1) RTG_RC(64, 7, 1)
#undef RTG_RC
  return RTG_EINVAL;
}
