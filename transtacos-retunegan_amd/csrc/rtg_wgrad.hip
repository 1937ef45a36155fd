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

#include "rtg_wgrad_kernel.h"

int rtg_wgrad_launch_m0(int, int, int, const rtg_wg::WgArgs&, dim3, size_t, hipStream_t);
int rtg_wgrad_launch_m1(int, int, int, const rtg_wg::WgArgs&, dim3, size_t, hipStream_t);
int rtg_wgrad_launch_m2(int, int, int, const rtg_wg::WgArgs&, dim3, size_t, hipStream_t);
int rtg_wgrad_launch_m3(int, int, int, const rtg_wg::WgArgs&, dim3, size_t, hipStream_t);
int rtg_wgrad_launch_m4(int, int, int, const rtg_wg::WgArgs&, dim3, size_t, hipStream_t);
int rtg_wgrad_launch_m5(int, int, int, const rtg_wg::WgArgs&, dim3, size_t, hipStream_t);
int rtg_wgrad_launch_m6(int, int, int, const rtg_wg::WgArgs&, dim3, size_t, hipStream_t);
int rtg_wgrad_launch_m7(int, int, int, const rtg_wg::WgArgs&, dim3, size_t, hipStream_t);

// rtg_wgrad_thin.hip: bandwidth kernels for the one-input-channel / one-output-channel shapes (shape code kThinShape)
int rtg_wgrad_thin_kind(const RtgWgradDesc* d);
int rtg_wgrad_thin_splits(const RtgWgradDesc* d);
int rtg_wgrad_thin_launch(const RtgWgradDesc* d, const float* x, const float* dy, const float* aux, float* part,
                          hipStream_t s);

// rtg_reswgrad.hip: streaming reduction for the stride-1 "same" convs of the UNet-G residual blocks (shape code kResShape)
int rtg_reswgrad_ok(const RtgWgradDesc* d);
int rtg_reswgrad_splits(const RtgWgradDesc* d);
int rtg_reswgrad_launch(const RtgWgradDesc* d, const float* x, const float* dy, float* part, hipStream_t s);

// rtg_gconv.hip: the thin-group k41 layers of MSD on the vector ALUs (shape code kGconvShape)
int rtg_gconv_wgrad_ok(const RtgWgradDesc* d);
int rtg_gconv_wgrad_splits(const RtgWgradDesc* d);
int rtg_gconv_wgrad_launch(const RtgWgradDesc* d, const float* x, const float* dy, float* part, hipStream_t s);

// rtg_gmfma.hip: the same layers on the matrix cores with exact-fit tiles (shape code kGmfmaShape)
int rtg_gmfma_wgrad_ok(const RtgWgradDesc* d);
int rtg_gmfma_wgrad_splits(const RtgWgradDesc* d);
int rtg_gmfma_wgrad_launch(const RtgWgradDesc* d, const float* x, const float* dy, float* part, hipStream_t s);

// rtg_dwgrad.hip: the dense discriminator layers with 16-byte operand fragments (shape code kDenseShape)
int rtg_dwgrad_variants(void);
int rtg_dwgrad_ok(const RtgWgradDesc* d, int variant);
int rtg_dwgrad_splits(const RtgWgradDesc* d, int variant);
int rtg_dwgrad_launch(const RtgWgradDesc* d, int variant, const float* x, const float* dy, float* part, hipStream_t s);

namespace {
using namespace rtg_wg;
constexpr int kThinShape = 7;
constexpr int kResShape = 8;
constexpr int kGconvShape = 9;
constexpr int kDenseShape = 10;       // 10 .. 10 + rtg_dwgrad_variants() - 1: its block shapes
constexpr int kDenseShapeLast = 14;
constexpr int kGmfmaShape = 15;

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
    if (d->shape_cfg && s != d->shape_cfg - 1) continue;        // caller-selected block shape
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
  if (o->shape < 0) return d->shape_cfg ? RTG_EINVAL : RTG_ERANGE;
  // one virtual sequence over all clips: seg_len slots per clip (its Q outputs + the gap that separates patches).  A
  // clip's patch is [left padding | L_in samples | right padding]; the right padding of one clip and the left padding of
  // the next are both zeros, so consecutive patches may OVERLAP by min(left, right) positions: the slots a clip's last
  // outputs read past its pitch are staged as the next clip's left padding (zero either way).  A "same" k5 conv on rows
  // of 10 takes 12 slots per clip instead of 14 (reduction steps spent on the gap: 17 % instead of 29 %).
  o->seg_pw = (d->Q - 1) * d->stride + (d->K - 1) * d->dil + 1;
  const int right = o->seg_pw - d->pad - d->L_in;            // zero slots after the clip's samples (< 0: samples unused)
  const int overlap = right > 0 ? (right < d->pad ? right : d->pad) : 0;
  int Lseg = (o->seg_pw - overlap + d->stride - 1) / d->stride;
  if (Lseg < d->Q) Lseg = d->Q;
  o->seg_len = Lseg;
  if ((long long)d->B * Lseg * d->stride + RTG_PW_MAX >= (1ll << 24)) return RTG_ERANGE;   // float-reciprocal division
  const int per_clip = rtg_ceil_div(d->Q, TT);
  o->cont = ((double)d->Q / Lseg > 1.08 * (double)d->Q / ((double)per_clip * TT) && d->B >= 2) ? 1 : 0;
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
  if (d->shape_cfg < 0 || (d->shape_cfg > kNumShapes && d->shape_cfg != kThinShape && d->shape_cfg != kResShape &&
                           d->shape_cfg != kGconvShape && d->shape_cfg != kGmfmaShape && !(d->shape_cfg >= kDenseShape && d->shape_cfg <= kDenseShapeLast)))
    return RTG_EINVAL;
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

}  // namespace

// bf16 tensors (RtgWgradDesc.io_bf16) with shape_cfg 0: the library's pick among the dense-layer kernel's shapes — the one
// with the most channel chunks per block that serves the problem (fewest operand loads per matrix instruction); 0: none
static int dense_default_shape(const RtgWgradDesc* d) {
  for (int v = 3; v >= 0; --v)
    if (rtg_dwgrad_ok(d, v)) return kDenseShape + v;
  for (int v = 4; v < rtg_dwgrad_variants(); ++v)
    if (rtg_dwgrad_ok(d, v)) return kDenseShape + v;
  return 0;
}

extern "C" int rtg_wgrad_splits(const RtgWgradDesc* d_in) {
  if (!d_in) return RTG_ENULL;
  RtgWgradDesc dd = *d_in;
  if (dd.io_bf16 != 0 && dd.shape_cfg == 0) dd.shape_cfg = dense_default_shape(&dd);
  const RtgWgradDesc* d = &dd;
  int st = validate(d);
  if (st) return st;
  if (d->io_bf16 != 0 && !(d->shape_cfg >= kDenseShape && d->shape_cfg <= kDenseShapeLast)) return RTG_ERANGE;
  if (d->shape_cfg == kThinShape) return rtg_wgrad_thin_splits(d);
  if (d->shape_cfg == kResShape) return rtg_reswgrad_splits(d);
  if (d->shape_cfg == kGconvShape) return rtg_gconv_wgrad_splits(d);
  if (d->shape_cfg == kGmfmaShape) return rtg_gmfma_wgrad_splits(d);
  if (d->shape_cfg >= kDenseShape && d->shape_cfg <= kDenseShapeLast) return rtg_dwgrad_splits(d, d->shape_cfg - kDenseShape);
  WgGeom g;
  st = geometry(d, &g);
  if (st) return st;
  const long long base = (long long)d->groups * g.m_blocks * g.n_cchunk;
  const long long total = g.n_tiles_total;
  // Cost model in microseconds: the matrix pipe of a CU is shared by its resident blocks, so the launch lasts about
  // (blocks per CU, rounded up) x (tiles per block, rounded up) tile times, plus a fixed cost per block and the
  // write + fixed-order read-back of one partial per split.
  const Shape sh = kShapes[g.shape];
  const double t_tile = sh.MTW * sh.NTW * (g.TM == 32 ? 0.98 : 0.25) + 0.35;
  const double t_fixed = 5.0;
  const double t_flush = (double)d->groups * d->Mg * ((double)d->Cg * d->K + 1) * 8.0 / 3.0e6;
  double best = 1e30;
  long long best_s = 1;
  const long long s_max = total < 512 ? total : 512;
  for (long long s = 1; s <= s_max; ++s) {
    const long long tiles = (total + s - 1) / s;
    const long long rounds = (base * s + 255) / 256;
    double t = (double)rounds * ((double)tiles * t_tile + t_fixed) + (double)s * t_flush;
    if (rounds == 1) t *= 1.15;                        // a lone block per CU cannot hide its own barrier phases
    if (t < best) { best = t; best_s = s; }
  }
  return (int)best_s;
}

extern "C" int rtg_wgrad_shape_candidates(const RtgWgradDesc* d, int* cfgs, int max) {
  if (!d || !cfgs) return RTG_ENULL;
  if (max < 1) return RTG_EINVAL;
  int st = validate(d);
  if (st) return st;
  int cnt = 0;
  if (d->io_bf16 != 0) {
    // bf16 tensors (ABI 9): the dense-layer kernel's shapes or nothing (the caller then converts around an fp32 launch)
    for (int v = 0; v < rtg_dwgrad_variants(); ++v)
      if (rtg_dwgrad_ok(d, v) && cnt < max) cfgs[cnt++] = kDenseShape + v;
    return cnt;
  }
  RtgWgradDesc t = *d;
  WgGeom g;
  t.shape_cfg = 0;
  st = geometry(&t, &g);
  if (st) return st;
  cfgs[cnt++] = g.shape + 1;                                    // the heuristic's choice first
  for (int s = 0; s < kNumShapes && cnt < max; ++s) {
    if (s == g.shape) continue;
    t.shape_cfg = s + 1;
    WgGeom gs;
    if (geometry(&t, &gs) == RTG_OK) cfgs[cnt++] = s + 1;
  }
  if (rtg_wgrad_thin_kind(d) > 0 && cnt < max) cfgs[cnt++] = kThinShape;
  if (rtg_reswgrad_ok(d) && cnt < max) cfgs[cnt++] = kResShape;
  if (rtg_gconv_wgrad_ok(d) && cnt < max) cfgs[cnt++] = kGconvShape;
  if (rtg_gmfma_wgrad_ok(d) && cnt < max) cfgs[cnt++] = kGmfmaShape;
  for (int v = 0; v < rtg_dwgrad_variants(); ++v)
    if (rtg_dwgrad_ok(d, v) && cnt < max) cfgs[cnt++] = kDenseShape + v;
  return cnt;
}

namespace {

struct WgPlan {
  WgArgs a;
  WgGeom g;
  unsigned blocks;
  size_t lds_bytes;
  int mode;                 // bit 0: continuous tiling, bit 1: second dimension, bit 2: bf16 operands
};

// checks + geometry + kernel arguments of one problem served by the general (matrix-core) kernel
int wgrad_plan(const RtgWgradDesc* d, const float* x1, const float* x2, const float* dy, const float* gy_aux,
               float* part, WgPlan* pl) {
  if (!d || !x1 || !dy || !part) return RTG_ENULL;
  int st = validate(d);
  if (st) return st;
  if (d->C2 > 0 && !x2) return RTG_ENULL;
  if ((d->gy_mode == RTG_PRE_MUL_DLRELU || d->gy_mode == RTG_PRE_MUL_DTANH) && !gy_aux) return RTG_ENULL;
  if (d->pre_mode != RTG_PRE_NONE && d->pre_mode != RTG_PRE_LRELU) return RTG_EINVAL;
  if (d->splits < 1 || d->splits > 65535) return RTG_EINVAL;
  if (d->io_bf16 != 0) return RTG_EINVAL;                  // (bf16 tensors: the dense-layer kernel only)
  const long long need = (long long)d->groups * d->Mg * ((long long)d->Cg * d->K + 1);
  if (d->splits > 1 && d->part_stride < need) return RTG_EINVAL;
  if (d->shape_cfg == kThinShape || d->shape_cfg == kResShape || d->shape_cfg == kGconvShape || d->shape_cfg == kGmfmaShape ||
      (d->shape_cfg >= kDenseShape && d->shape_cfg <= kDenseShapeLast))
    return RTG_EINVAL;                                                               // kernels of their own
  WgGeom& g = pl->g;
  st = geometry(d, &g);
  if (st) return st;
  const Shape sh = kShapes[g.shape];

  WgArgs& a = pl->a;
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
  pl->lds_bytes = (size_t)(a.ones_off + TT * d->stride + 8) * sizeof(float);
  const long long gy = (long long)d->groups * g.m_blocks * g.n_cchunk;
  const long long n_items = gy * d->splits;
  if (n_items > (1ll << 28)) return RTG_ERANGE;
  a.gy = (int)gy; a.n_items = (int)n_items;
  a.per_xcd = (int)((n_items + 7) / 8);
  pl->blocks = (unsigned)(8 * a.per_xcd);
  pl->mode = (g.cont ? 1 : 0) | (two_d ? 2 : 0) | (d->bf16 ? 4 : 0);
  return RTG_OK;
}

}  // namespace

int rtg_wgrad_launch_group_m0(int, const rtg_wg::WgGroupArgs&, size_t, hipStream_t);
int rtg_wgrad_launch_group_m1(int, const rtg_wg::WgGroupArgs&, size_t, hipStream_t);

extern "C" int rtg_conv1d_wgrad(const RtgWgradDesc* d_in, const float* x1, const float* x2, const float* dy,
                                const float* gy_aux, float* part, void* stream) {
  if (!d_in || !x1 || !dy || !part) return RTG_ENULL;
  RtgWgradDesc dd = *d_in;
  if (dd.io_bf16 != 0 && dd.shape_cfg == 0) dd.shape_cfg = dense_default_shape(&dd);      // (as rtg_wgrad_splits picked)
  const RtgWgradDesc* d = &dd;
  if (d->shape_cfg == kThinShape || d->shape_cfg == kResShape || d->shape_cfg == kGconvShape || d->shape_cfg == kGmfmaShape ||
      (d->shape_cfg >= kDenseShape && d->shape_cfg <= kDenseShapeLast)) {
    int st = validate(d);
    if (st) return st;
    if ((d->gy_mode == RTG_PRE_MUL_DLRELU || d->gy_mode == RTG_PRE_MUL_DTANH) && !gy_aux) return RTG_ENULL;
    if (d->pre_mode != RTG_PRE_NONE && d->pre_mode != RTG_PRE_LRELU) return RTG_EINVAL;
    if (d->splits < 1 || d->splits > 65535) return RTG_EINVAL;
    const long long need = (long long)d->groups * d->Mg * ((long long)d->Cg * d->K + 1);
    if (d->splits > 1 && d->part_stride < need) return RTG_EINVAL;
    if (d->io_bf16 != 0 && !(d->shape_cfg >= kDenseShape && d->shape_cfg <= kDenseShapeLast)) return RTG_EINVAL;
    if (d->shape_cfg == kThinShape) return rtg_wgrad_thin_launch(d, x1, dy, gy_aux, part, (hipStream_t)stream);
    if (d->shape_cfg == kGconvShape) return rtg_gconv_wgrad_launch(d, x1, dy, part, (hipStream_t)stream);
    if (d->shape_cfg == kGmfmaShape) return rtg_gmfma_wgrad_launch(d, x1, dy, part, (hipStream_t)stream);
    if (d->shape_cfg >= kDenseShape && d->shape_cfg <= kDenseShapeLast)
      return rtg_dwgrad_launch(d, d->shape_cfg - kDenseShape, x1, dy, part, (hipStream_t)stream);
    return rtg_reswgrad_launch(d, x1, dy, part, (hipStream_t)stream);
  }
  WgPlan pl;
  const int st = wgrad_plan(d, x1, x2, dy, gy_aux, part, &pl);
  if (st) return st;
  const WgGeom& g = pl.g;
  const WgArgs& a = pl.a;
  const dim3 grid(pl.blocks, 1, 1);
  const size_t lds_bytes = pl.lds_bytes;
  hipStream_t s = (hipStream_t)stream;
  switch (pl.mode) {
    case 0: return rtg_wgrad_launch_m0(g.TM, g.shape, g.maxit, a, grid, lds_bytes, s);
    case 1: return rtg_wgrad_launch_m1(g.TM, g.shape, g.maxit, a, grid, lds_bytes, s);
    case 2: return rtg_wgrad_launch_m2(g.TM, g.shape, g.maxit, a, grid, lds_bytes, s);
    case 3: return rtg_wgrad_launch_m3(g.TM, g.shape, g.maxit, a, grid, lds_bytes, s);
    case 4: return rtg_wgrad_launch_m4(g.TM, g.shape, g.maxit, a, grid, lds_bytes, s);
    case 5: return rtg_wgrad_launch_m5(g.TM, g.shape, g.maxit, a, grid, lds_bytes, s);
    case 6: return rtg_wgrad_launch_m6(g.TM, g.shape, g.maxit, a, grid, lds_bytes, s);
    default: return rtg_wgrad_launch_m7(g.TM, g.shape, g.maxit, a, grid, lds_bytes, s);
  }
}

// n problems in ONE launch.  Every member names the same general block shape (shape_cfg 1..6: the caller fixes it, as
// for rtg_conv1d_group) and must map to the same kernel instance: fp32 operands, 1-D rows, 32-row MFMA tiles (Mg >= 32),
// the same tiling mode and a staged patch of at most 128 floats per channel.  RTG_EINVAL otherwise — the caller then
// launches the members one by one.  Results are bit-identical to the members' own launches.
extern "C" int rtg_conv1d_wgrad_group(int n, const RtgWgradDesc* descs, const RtgWgradPtrs* ptrs, void* stream) {
  if (!descs || !ptrs) return RTG_ENULL;
  if (n < 1 || n > RTG_WG_MAX_GROUP) return RTG_EINVAL;
  rtg_wg::WgGroupArgs ga;
  ga.n = n;
  size_t lds_bytes = 0;
  unsigned end = 0;
  int shape = -1, mode = -1;
  for (int i = 0; i < n; ++i) {
    if (descs[i].shape_cfg < 1 || descs[i].shape_cfg > kNumShapes) return RTG_EINVAL;
    WgPlan pl;
    const RtgWgradPtrs& q = ptrs[i];
    const int st = wgrad_plan(&descs[i], q.x1, q.x2, q.dy, q.gy_aux, q.part, &pl);
    if (st) return st;
    if (pl.g.TM != 32 || pl.g.maxit != 2 || (pl.mode & ~1) != 0) return RTG_EINVAL;
    if (i == 0) { shape = pl.g.shape; mode = pl.mode; }
    else if (pl.g.shape != shape || pl.mode != mode) return RTG_EINVAL;
    ga.p[i] = pl.a;
    end += pl.blocks;
    ga.blk_end[i] = end;
    lds_bytes = pl.lds_bytes > lds_bytes ? pl.lds_bytes : lds_bytes;
  }
  for (int i = n; i < RTG_WG_MAX_GROUP; ++i) ga.blk_end[i] = end;
  hipStream_t s = (hipStream_t)stream;
  return mode == 0 ? rtg_wgrad_launch_group_m0(shape, ga, lds_bytes, s) : rtg_wgrad_launch_group_m1(shape, ga, lds_bytes, s);
}
