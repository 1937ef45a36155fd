// rtg_conv1d.hip — implicit-GEMM Conv1d on the gfx950 fp32 matrix cores.
//
// One kernel family serves: Conv1d forward (any stride / dilation / groups), ConvTranspose1d forward and every
// convolution backward-data of the RetuneGAN path (through repacked weights: rtg_weights.hip).  See include/rtg.h
// (rtg_conv1d) for the operator definition and the reference call sites it replaces.
//
// Mapping to the hardware:
//   GEMM view   M = output rows of one group, N = batch x output positions, K = (input channel, tap).
//   workgroup   256 threads = 4 wavefronts arranged WM x WN; wave tile = MT x NT MFMA tiles of TM x TM
//               (TM = 32: v_mfma_f32_32x32x2_f32, TM = 16: v_mfma_f32_16x16x4_f32 for 16-row groups).
//   B operand   the input patch of 16 channels x (BN-1)*stride+(K-1)*dil+1 samples is staged ONCE per block into LDS
//               (coalesced, branch-free global loads issued back to back; input activation fused at staging), double
//               buffered across channel chunks; every tap / every output row re-reads it from LDS with ds_read_b32 at
//               lane-consecutive addresses (strided convs de-interleave the patch by phase: bank-conflict free).
//   short rows  when a clip's output row is much shorter than the block tile (MPD / MSD tails: 10..128 positions)
//               several clips are packed into one tile: the columns enumerate (clip, q) densely, the staged patch
//               gives every clip a "segment" of seg_len virtual positions (outputs + halo gap), so the MFMA columns
//               are spent neither on padding nor on the halo.
//   A operand   weights are pre-packed (rtg_weights_pack) so that one MFMA fragment is 64 consecutive floats:
//               each wave loads its fragments straight from L2 with one coalesced 256-B load, prefetched one
//               (chunk, tap) step ahead — no LDS traffic and no barrier for weights.
//   epilogue    bias, leaky-relu-derivative mask, residual, scale, activation, optional polyphase "shuffle" store.
// The kernel lives in rtg_conv1d_kernel.h; its block-shape instances are compiled in rtg_conv1d_t*.hip.
#include "rtg_conv1d_kernel.h"


int rtg_conv1d_launch_group_32_1_1(const rtg_cv::GroupArgs&, size_t, int, hipStream_t);
int rtg_conv1d_launch_group_32_1_2(const rtg_cv::GroupArgs&, size_t, int, hipStream_t);
int rtg_conv1d_launch_group_32_1_4(const rtg_cv::GroupArgs&, size_t, int, hipStream_t);
int rtg_conv1d_launch_group_32_2_1(const rtg_cv::GroupArgs&, size_t, int, hipStream_t);
int rtg_conv1d_launch_group_32_2_2(const rtg_cv::GroupArgs&, size_t, int, hipStream_t);
int rtg_conv1d_launch_group_16_1_1(const rtg_cv::GroupArgs&, size_t, int, hipStream_t);
int rtg_conv1d_launch_group_16_1_2(const rtg_cv::GroupArgs&, size_t, int, hipStream_t);
int rtg_conv1d_launch_group_16_1_4(const rtg_cv::GroupArgs&, size_t, int, hipStream_t);

using rtg_cv::ConvArgs;

#ifdef RTG_STAMPS
// diagnostic builds only: a 64 MB device buffer of per-(block, wave) s_memtime stamps, read back by the dev tools
static unsigned long long* rtg_dev_stamps() {
  static unsigned long long* p = nullptr;
  if (!p && hipMalloc((void**)&p, 8u << 23) != hipSuccess) p = nullptr;
  return p;
}
extern "C" int rtg_dev_stamp_read(unsigned long long* host, long long n) {
  return hipMemcpy(host, rtg_dev_stamps(), n * 8, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
extern "C" int rtg_dev_stamp_clear() { return hipMemset(rtg_dev_stamps(), 0, 8u << 23) == hipSuccess ? 0 : -1; }
#endif

namespace {

struct TileCfg {
  int MT, NT, WM, WN;
  int seg_len, seg_nb;   // > 0: clips packed per block
  int dense;             // packed, and the block's columns are a window of the dense (clip, q) sequence: clips may
                         // straddle blocks (seg_nb = segments a block stages, partial clips included)
};

// Patch width (floats per channel) for a block covering BN output positions.
inline int patch_width(int BN, int stride, int K, int dil) { return (BN - 1) * stride + (K - 1) * dil + 1; }

// virtual positions one clip occupies when clips are packed side by side: its Q outputs plus the gap that keeps the
// next clip's input patch from overlapping (pitch = seg_len * stride >= (Q-1)*stride + (K-1)*dil + 1)
inline int segment_len(int Q, int stride, int K, int dil) {
  const int extra = (K - 1) * dil + 1 - stride;
  return Q + (extra > 0 ? (extra + stride - 1) / stride : 0);
}

// Scores every block shape valid for the problem (score <= 0: not applicable); returns the number of shapes.
constexpr int kMaxTileCfgs = 32;
struct ScoredCfg {
  TileCfg c;
  double score;
};
inline int tile_code(const TileCfg& c) { return c.dense * 1000 + c.MT * 100 + c.NT * 10 + c.WM; }

int score_tiles(int TM, int n_mt, int Q, int B, int groups, int stride, int K, int dil, bool windows, ScoredCfg* out) {
  static const int c32[][4] = {{2, 2, 1, 4}, {2, 2, 2, 2}, {2, 2, 4, 1}, {1, 2, 1, 4}, {1, 2, 2, 2}, {1, 2, 4, 1},
                               {1, 4, 1, 4}, {1, 4, 2, 2}, {2, 1, 1, 4}, {2, 1, 2, 2}, {2, 1, 4, 1}, {1, 1, 1, 4},
                               {1, 1, 2, 2}, {1, 1, 4, 1}};
  static const int c16[][4] = {{1, 4, 1, 4}, {1, 2, 1, 4}, {1, 1, 1, 4}, {1, 4, 2, 2}, {1, 2, 2, 2}, {1, 1, 2, 2},
                               {1, 2, 4, 1}, {1, 1, 4, 1}};
  const int(*cs)[4] = TM == 32 ? c32 : c16;
  const int n = TM == 32 ? (int)(sizeof(c32) / sizeof(c32[0])) : (int)(sizeof(c16) / sizeof(c16[0]));
  const int Lseg = segment_len(Q, stride, K, dil);
  int cnt = 0;
  for (int i = 0; i < n; ++i) {
    TileCfg c = {cs[i][0], cs[i][1], cs[i][2], cs[i][3], 0, 0, 0};
    const int BN = c.WN * c.NT * TM;
    const int mrows = c.WM * c.MT;
    const int m_blocks = rtg_ceil_div(n_mt, mrows);
    double eff_q, n_blocks_q;
    // clips packed per block: columns enumerate (clip, q) densely (Q per clip), the staged patch holds Lseg virtual
    // positions per clip
    int nb = BN / Q;
    while (nb >= 2 && patch_width(nb * Lseg, stride, K, dil) > RTG_PW_MAX) --nb;
    if ((nb < 2 || B < 2) && patch_width(BN, stride, K, dil) > RTG_PW_MAX) continue;
    if (nb >= 2 && B >= 2) {
      c.seg_len = Lseg;
      c.seg_nb = nb < B ? nb : B;
      const int zb = rtg_ceil_div(B, c.seg_nb);
      eff_q = ((double)B * Q) / ((double)zb * BN);
      n_blocks_q = zb;
    } else {
      const int q_blocks = rtg_ceil_div(Q, BN);
      eff_q = (double)Q / ((double)q_blocks * BN);
      n_blocks_q = (double)q_blocks * B;
    }
    const double eff = ((double)n_mt / (m_blocks * mrows)) * eff_q;
    const double blocks = (double)m_blocks * n_blocks_q * groups;
    const double fill = blocks >= 512.0 ? 1.0 : blocks / 512.0;
    const double reuse = (double)(c.MT * c.NT) / (c.MT + c.NT);   // MFMAs per operand fragment fetched
    out[cnt].c = c;
    out[cnt].score = eff * fill * (0.6 + 0.4 * (reuse > 1.0 ? 1.0 : reuse)) + 1e-9;
    ++cnt;
    // the same shape over the dense (clip, q) column sequence: no column is lost to whole-clip quantisation, a block
    // stages every clip its window touches (not for the class-pure blocks of strided 2-D backward-data; rows shorter
    // than the tile; the wider patch must still fit)
    if (windows && B >= 2 && Q < BN && cnt < kMaxTileCfgs) {
      const int segs0 = (Q - 1 + BN - 1) / Q + 1;
      const int segs = segs0 < B ? segs0 : B;
      const long long cols = (long long)B * Q;
      const long long zb = (cols + BN - 1) / BN;
      const double eff_d = (double)cols / ((double)zb * BN);
      if (patch_width(segs * Lseg, stride, K, dil) <= RTG_PW_MAX && eff_d > eff_q + 0.02) {
        TileCfg cd = c;
        cd.dense = 1; cd.seg_len = Lseg; cd.seg_nb = segs;
        const double blocks_d = (double)m_blocks * zb * groups;
        const double fill_d = blocks_d >= 512.0 ? 1.0 : blocks_d / 512.0;
        out[cnt].c = cd;
        out[cnt].score = ((double)n_mt / (m_blocks * mrows)) * eff_d * fill_d * (0.6 + 0.4 * (reuse > 1.0 ? 1.0 : reuse)) *
                             0.97 + 1e-9;      // more staging per block than the whole-clip packing of equal efficiency
        ++cnt;
      }
    }
  }
  return cnt;
}

// forced: a RtgConv1dDesc.tile_cfg code, or 0 for the best score.  MT == 0 in the result: nothing applicable.
TileCfg pick_tiles(int TM, int n_mt, int Q, int B, int groups, int stride, int K, int dil, bool windows, int forced) {
  ScoredCfg sc[kMaxTileCfgs];
  const int n = score_tiles(TM, n_mt, Q, B, groups, stride, K, dil, windows, sc);
  TileCfg best = {0, 0, 0, 0, 0, 0, 0};
  double best_score = -1.0;
  for (int i = 0; i < n; ++i) {
    if (forced ? tile_code(sc[i].c) == forced : sc[i].score > best_score) {
      best = sc[i].c;
      best_score = sc[i].score;
      if (forced) break;
    }
  }
  return best;
}

}  // namespace

// rtg_thin.hip: bandwidth kernels for the one-input-channel / one-output-channel shapes
int rtg_thin_kind(const RtgConv1dDesc* d);
int rtg_thin_launch(int kind, const RtgConv1dDesc* d, const float* x, const float* aux, const float* wp,
                    const float* bias, const float* mask, const float* res, float* out, hipStream_t s);

// rtg_resconv.hip: weights-in-registers kernel for the stride-1 "same" convolutions of the UNet-G residual blocks
// (block-shape codes 7001 / 7002 = 32 / 64 positions per wave)
int rtg_resconv_variants(const RtgConv1dDesc* d);
int rtg_resconv_launch(const RtgConv1dDesc* d, int nt, const float* x, const float* wp, const float* bias,
                       const float* mask, const float* res, float* out, hipStream_t s);
#define RTG_RESCONV_CODE 7000

// rtg_dconv.hip: the dense-layer kernel (16-byte operand fragments, 16-column tiles) for the 128..512-channel layers of
// the discriminators (block-shape codes 8000 + 100 * shape + 16-column tiles per block; needs RtgConv1dDesc.wp16)
int rtg_dconv_candidates(const RtgConv1dDesc* d, int* codes, int max);
int rtg_dconv_launch(const RtgConv1dDesc* d, int code, const float* x, const float* wp, const float* bias,
                     const float* mask, const float* res, float* out, hipStream_t s);
#define RTG_DCONV_CODE 8000
// rtg_sconv.hip: stride-1 "same" convolutions over few columns (the bottom of the UNet) with split-K over the waves of a
// block (block-shape codes 9000 + waves; needs RtgConv1dDesc.wp16)
int rtg_sconv_candidates(const RtgConv1dDesc* d, int* codes, int max);
int rtg_sconv_launch(const RtgConv1dDesc* d, int code, const float* x1, const float* x2, const float* wp, const float* bias,
                     const float* mask, const float* res, float* out, float* out2, hipStream_t s);
#define RTG_SCONV_CODE 9000
static bool sconv_code_ok(const RtgConv1dDesc* d) {
  int codes[4];
  const int n = rtg_sconv_candidates(d, codes, 4);
  for (int i = 0; i < n; ++i)
    if (codes[i] == d->tile_cfg) return true;
  return false;
}
// descriptors only the dense-layer kernel serves: bf16 tensors (io_bf16), the 2-D forward over a kernel-row-major image (h_mode 2)
static bool dense_only(const RtgConv1dDesc* d) {
  return d->io_bf16 != 0 || ((d->h_k > 1 || d->h_n > 1) && d->h_mode == 2);
}

static bool dconv_code_ok(const RtgConv1dDesc* d) {
  int codes[24];
  const int n = rtg_dconv_candidates(d, codes, 24);
  for (int i = 0; i < n; ++i)
    if (codes[i] == d->tile_cfg) return true;
  return false;
}

extern "C" int rtg_conv1d_variant(const RtgConv1dDesc* d) {
  if (!d) return RTG_ENULL;
  if (dense_only(d)) {
    int code = d->tile_cfg;
    if (code == 0 && rtg_dconv_candidates(d, &code, 1) < 1) return RTG_ERANGE;
    // (a caller-fixed shape must be one the candidate list holds: the list leaves out what must not run — instances that
    // spill, 256-row blocks that do not divide the rows, 64-row waves without bf16 — and a stale tuner table must not get past it)
    if (d->tile_cfg != 0 && !dconv_code_ok(d)) return RTG_EINVAL;
    return (code > RTG_DCONV_CODE && code <= RTG_SCONV_CODE) ? code : RTG_EINVAL;
  }
  if (d->tile_cfg > RTG_SCONV_CODE) return sconv_code_ok(d) ? d->tile_cfg : RTG_EINVAL;
  if (d->tile_cfg > RTG_DCONV_CODE) return dconv_code_ok(d) ? d->tile_cfg : RTG_EINVAL;
  if (d->tile_cfg > RTG_RESCONV_CODE && d->tile_cfg <= RTG_RESCONV_CODE + 2)
    return (rtg_resconv_variants(d) & (1 << (d->tile_cfg - RTG_RESCONV_CODE - 1))) ? d->tile_cfg : RTG_EINVAL;
  if ((d->tile_m != 32 && d->tile_m != 16) || d->Mg < 1 || d->Q < 1 || d->B < 1 || d->groups < 1 || d->stride < 1 ||
      d->K < 1 || d->dil < 1)
    return RTG_EINVAL;
  if (d->tile_cfg == 0 && rtg_thin_kind(d)) return rtg_thin_kind(d);
  const TileCfg c = pick_tiles(d->tile_m, rtg_ceil_div(d->Mg, d->tile_m), d->Q, d->B, d->groups, d->stride, d->K, d->dil,
                               !((d->h_k > 1 || d->h_n > 1) && d->h_mode == 1 && d->h_stride > 1),
                               d->tile_cfg);
  if (c.MT == 0) return d->tile_cfg ? RTG_EINVAL : RTG_ERANGE;
  return d->tile_m * 100 + c.MT * 10 + c.NT;
}

extern "C" int rtg_conv1d_tile_candidates(const RtgConv1dDesc* d, int* cfgs, int max) {
  if (!d || !cfgs) return RTG_ENULL;
  if ((d->tile_m != 32 && d->tile_m != 16) || d->Mg < 1 || d->Q < 1 || d->B < 1 || d->groups < 1 || d->stride < 1 ||
      d->K < 1 || d->dil < 1 || max < 1)
    return RTG_EINVAL;
  if (dense_only(d)) {
    // bf16 tensors (ABI 9): only the dense-layer kernel reads / writes them — its shapes or nothing (the caller converts)
    return rtg_dconv_candidates(d, cfgs, max < 10 ? max : 10);
  }
  if (rtg_thin_kind(d)) {       // served by a bandwidth kernel: nothing to choose (0 = the library's default)
    cfgs[0] = 0;
    return 1;
  }
  ScoredCfg sc[kMaxTileCfgs];
  const int n = score_tiles(d->tile_m, rtg_ceil_div(d->Mg, d->tile_m), d->Q, d->B, d->groups, d->stride, d->K, d->dil,
                            !((d->h_k > 1 || d->h_n > 1) && d->h_mode == 1 && d->h_stride > 1), sc);
  // best-guess first (selection sort by score; n <= 14)
  int cnt = 0;
  for (int k = 0; k < n && cnt < max; ++k) {
    int bi = -1;
    for (int i = 0; i < n; ++i)
      if (sc[i].score > 0.0 && (bi < 0 || sc[i].score > sc[bi].score)) bi = i;
    if (bi < 0) break;
    cfgs[cnt++] = tile_code(sc[bi].c);
    sc[bi].score = -1.0;
  }
  const int rv = rtg_resconv_variants(d);
  for (int nt = 2; nt >= 1; --nt)
    if ((rv & (1 << (nt - 1))) && cnt < max) cfgs[cnt++] = RTG_RESCONV_CODE + nt;
  // the dense-layer kernel's best-scored shapes (all of them would double the tuning step's work on these layers)
  if (cnt < max) cnt += rtg_dconv_candidates(d, cfgs + cnt, max - cnt < 10 ? max - cnt : 10);
  if (cnt < max) cnt += rtg_sconv_candidates(d, cfgs + cnt, max - cnt);
  return cnt;
}

extern "C" long long rtg_packed_size(int groups, int Mg, int Cg, int K, int tile_m) {
  if (groups < 1 || Mg < 1 || Cg < 1 || K < 1 || (tile_m != 32 && tile_m != 16)) return RTG_EINVAL;
  const long long n_mt = (Mg + tile_m - 1) / tile_m, n_cc = (Cg + RTG_CK - 1) / RTG_CK;
  return (long long)groups * n_mt * n_cc * K * RTG_CK * tile_m;
}

extern "C" long long rtg_packed_size_bf16(int groups, int Mg, int Cg, int K, int tile_m) {
  if (groups < 1 || Mg < 1 || Cg < 1 || K < 1 || (tile_m != 32 && tile_m != 16)) return RTG_EINVAL;
  const long long n_mt = (Mg + tile_m - 1) / tile_m, n_cc = (Cg + RTG_CK - 1) / RTG_CK;
  return (long long)groups * n_mt * n_cc * K * (tile_m == 32 ? 2 : 1) * 64 * 2;     // 8 bytes per lane and MFMA
}

// k-step groups of the tap-major order: Cg * ceil(K / KK) k-steps in groups of CPN
static int tapmajor_groups(int Cg, int K, int tile_m) {
  const int KK = 64 / tile_m, CPN = RTG_CK / KK;
  return rtg_ceil_div(Cg * rtg_ceil_div(K, KK), CPN);
}

extern "C" int rtg_tapmajor_pays(int Cg, int K, int tile_m) {
  if (Cg < 1 || K < 1 || (tile_m != 32 && tile_m != 16)) return RTG_EINVAL;
  if (Cg > RTG_CK) return 0;
  return tapmajor_groups(Cg, K, tile_m) < K ? 1 : 0;       // channel-major needs K groups of CPN k-steps per chunk
}

extern "C" long long rtg_packed_size_tapmajor(int groups, int Mg, int Cg, int K, int tile_m) {
  if (groups < 1 || Mg < 1 || Cg < 1 || Cg > RTG_CK || K < 1 || (tile_m != 32 && tile_m != 16)) return RTG_EINVAL;
  const long long n_mt = (Mg + tile_m - 1) / tile_m;
  return (long long)groups * n_mt * tapmajor_groups(Cg, K, tile_m) * RTG_CK * tile_m;
}

// One problem's kernel arguments, grid and LDS size from its descriptor (shared by the single and the grouped launch).
// thin_ok: may be served by a bandwidth kernel (rtg_thin.hip) -> returns 1 with *thin set.
struct ConvPlan {
  ConvArgs a;
  unsigned blocks;
  size_t lds_bytes;
  int TM, MT, NT, bf;
};

static int conv_plan(const RtgConv1dDesc* d, const float* x1, const float* x2, const float* aux, const float* wp,
                     const float* bias, const float* mask, const float* res, float* out, float* out2, ConvPlan* pl) {
  if (!d || !x1 || !wp) return RTG_ENULL;
  if (dense_only(d)) return RTG_EINVAL;                    // (the dense-layer kernel only, rtg_conv1d routes them)
  if (d->out_split == 0 ? !out : (!out && !out2)) return RTG_ENULL;
  if (d->out_split < 0 || d->out_split >= d->out_C) return RTG_EINVAL;
  if (d->out_split > 0 && (mask || res)) return RTG_EINVAL;
  if (d->B < 1 || d->C1 < 1 || d->C2 < 0 || d->L_in < 1 || d->groups < 1 || d->Cg < 1 || d->Mg < 1 || d->K < 1 ||
      d->stride < 1 || d->dil < 1 || d->Q < 1 || d->out_C < 1 || d->out_L < 1 || d->shuf_S < 1)
    return RTG_EINVAL;
  if (d->tile_m != 32 && d->tile_m != 16) return RTG_EINVAL;
  if (d->C1 + d->C2 != d->groups * d->Cg) return RTG_EINVAL;
  if (d->C2 > 0 && !x2) return RTG_ENULL;
  if ((d->pre_mode == RTG_PRE_MUL_DLRELU || d->pre_mode == RTG_PRE_MUL_DTANH) && (!aux || d->C2 != 0)) return RTG_EINVAL;
  if (d->stride > 1 && d->dil != 1) return RTG_ERANGE;
  if ((long long)d->groups * d->Mg != (long long)d->out_C * d->shuf_S) return RTG_EINVAL;
  if (d->shuf_S == 1 && d->Q > d->out_L) return RTG_EINVAL;
  // second dimension (all zero / one = plain 1-D)
  const bool two_d = d->h_k > 1 || d->h_n > 1;
  if (two_d) {
    if (d->h_in < 1 || d->h_k < 1 || d->h_stride < 1 || d->h_pad < 0 || d->h_n < 1 || (d->h_mode != 0 && d->h_mode != 1))
      return RTG_EINVAL;
    if (d->groups != 1 || d->C2 != 0 || d->out_split != 0 || d->C1 % d->h_k != 0 || d->B % d->h_n != 0) return RTG_EINVAL;
  }
  const long long x_bytes = two_d ? (long long)(d->B / d->h_n) * (d->C1 / d->h_k) * d->h_in * d->L_in * 4
                                  : (long long)d->B * d->C1 * d->L_in * 4;
  if (x_bytes >= (1ll << 31) || (long long)d->B * d->C2 * d->L_in * 4 >= (1ll << 31)) return RTG_ERANGE;   // 32-bit offsets
  if ((long long)(two_d ? d->B / d->h_n : d->B) * d->out_C * (two_d ? d->h_n : 1) * d->out_L * 4 >= (1ll << 31)) return RTG_ERANGE;

  ConvArgs& a = pl->a;
  a.x1 = x1; a.x2 = x2; a.aux = (d->pre_mode >= RTG_PRE_MUL_DLRELU) ? aux : nullptr; a.wp = wp; a.bias = bias;
  a.mask = mask; a.res = res; a.out = out; a.out2 = out2; a.out_split = d->out_split;
  a.B = d->B; a.C1 = d->C1; a.C2 = d->C2; a.L_in = d->L_in; a.groups = d->groups; a.Cg = d->Cg; a.Mg = d->Mg;
  a.K = d->K; a.stride = d->stride; a.dil = d->dil; a.pad = d->pad; a.Q = d->Q; a.out_C = d->out_C; a.out_L = d->out_L;
  a.shuf_S = d->shuf_S; a.shuf_P = d->shuf_P; a.pre_mode = d->pre_mode; a.pre_slope = d->pre_slope;
  a.mask_slope = d->mask_slope; a.out_scale = d->out_scale; a.act = d->act; a.act_slope = d->act_slope;
  a.accumulate = d->accumulate;
  a.two_d = two_d ? 1 : 0;
  a.h_in = two_d ? d->h_in : 1; a.h_k = two_d ? d->h_k : 1; a.h_stride = two_d ? d->h_stride : 1;
  a.h_pad = two_d ? d->h_pad : 0; a.h_n = two_d ? d->h_n : 1; a.h_mode = two_d ? d->h_mode : 0;
  a.x_bytes = (int)x_bytes; a.aux_bytes = a.aux ? (int)x_bytes : 0;
  const int TM = d->tile_m;
  a.n_cc = rtg_ceil_div(d->Cg, RTG_CK);
  a.n_mt = rtg_ceil_div(d->Mg, TM);
  a.tapmajor = d->tap_major ? 1 : 0;
  a.K_real = d->K;
  a.TG = rtg_ceil_div(d->K, 64 / TM);
  if (a.tapmajor) {
    if (d->Cg > RTG_CK) return RTG_EINVAL;
    a.K = tapmajor_groups(d->Cg, d->K, TM);     // the kernel's step loop walks groups of CPN k-steps
  }

  const TileCfg c = pick_tiles(TM, a.n_mt, d->Q, d->B, d->groups, d->stride, d->K, d->dil,
                               !(two_d && d->h_mode == 1 && d->h_stride > 1), d->tile_cfg);
  if (c.MT == 0) return d->tile_cfg ? RTG_EINVAL : RTG_ERANGE;   // unknown shape / even the smallest patch exceeds RTG_PW_MAX
  a.WM = c.WM; a.WN = c.WN;
  const int BN = c.WN * c.NT * TM;
  a.PW = patch_width(c.seg_len > 0 ? c.seg_nb * c.seg_len : BN, d->stride, d->K, d->dil);
  a.seg_len = c.seg_len; a.seg_nb = c.seg_nb; a.seg_dense = c.dense;
  a.seg_pitch = c.seg_len > 0 ? c.seg_len * d->stride : 1;
  a.seg_pw = patch_width(d->Q, d->stride, d->K, d->dil);
  int row;
  if (d->stride == 1) {
    a.PH = a.PW;
    row = a.PW;
  } else {
    a.PH = rtg_ceil_div(a.PW, d->stride) | 1;   // odd phase pitch
    row = a.PH * d->stride;
  }
  a.ROW = ((row + 15) / 32) * 32 + 16;          // == 16 (mod 32), >= row
  a.m_blocks = rtg_ceil_div(a.n_mt, c.WM * c.MT);
  int gz = c.seg_len > 0 ? rtg_ceil_div(d->B, c.seg_nb) : d->B;
  if (c.dense) gz = (int)(((long long)d->B * d->Q + BN - 1) / BN);
  if (two_d && d->h_mode == 1 && d->h_stride > 1 && c.seg_len > 0) {
    // class-pure blocks of packed rows (rtg_conv1d_kernel.h, RowClass): per batch item, per residue class of the rows,
    // ceil(rows of the class / seg_nb) blocks
    int per_item = 0;
    for (int r = 0; r < d->h_stride; ++r) {
      int f = (r - d->h_pad) % d->h_stride;
      if (f < 0) f += d->h_stride;
      per_item += f < d->h_n ? rtg_ceil_div(rtg_ceil_div(d->h_n - f, d->h_stride), c.seg_nb) : 0;
    }
    gz = (d->B / d->h_n) * per_item;
  }
  a.gx = c.seg_len > 0 ? 1 : rtg_ceil_div(d->Q, BN);
  const long long total = (long long)d->groups * a.m_blocks * a.gx * gz;
  if (total > (1ll << 30)) return RTG_ERANGE;
  a.total = (int)total;
  a.per_xcd = rtg_ceil_div(total, 8);
  pl->blocks = (unsigned)(8 * a.per_xcd);
#ifdef RTG_STAMPS
  a.dbg = rtg_dev_stamps();
#endif
  a.tab_off = 2 * RTG_CK * a.ROW;
  size_t lds_bytes = (size_t)(2 * RTG_CK * a.ROW + (a.tapmajor ? a.K * RTG_CK : 0)) * sizeof(float);
#ifdef RTG_STAMPS
  if (RTG_ENV_SET("RTG_DEV_OCC")) {            // diagnostic builds: cap the resident blocks per CU through LDS
    const int cap = RTG_ENV_INT("RTG_DEV_OCC", 0);
    if (cap > 0 && lds_bytes < (size_t)(160 * 1024 / cap - 1024)) lds_bytes = 160 * 1024 / cap - 1024;
  }
#endif
  pl->lds_bytes = lds_bytes;
  pl->TM = TM; pl->MT = c.MT; pl->NT = c.NT; pl->bf = d->bf16 ? 1 : 0;
  if (d->bf16 && (d->tap_major || (d->bf16 != 1))) return RTG_EINVAL;
  return RTG_OK;
}

extern "C" int rtg_conv1d(const RtgConv1dDesc* d, const float* x1, const float* x2, const float* aux, const float* wp,
                          const float* bias, const float* mask, const float* res, float* out, float* out2,
                          void* stream) {
  if (!d) return RTG_ENULL;
  if (dense_only(d)) {
    // bf16 tensors: the dense-layer kernel or nothing.  tile_cfg 0 (the library's heuristic) = its best-scored shape
    int code = d->tile_cfg;
    if (code == 0 && rtg_dconv_candidates(d, &code, 1) < 1) return RTG_ERANGE;
    if (code <= RTG_DCONV_CODE || code > RTG_SCONV_CODE || x2 || aux || out2) return RTG_EINVAL;
    if (d->tile_cfg != 0 && !dconv_code_ok(d)) return RTG_EINVAL;
    return rtg_dconv_launch(d, code, x1, wp, bias, mask, res, out, (hipStream_t)stream);
  }
  if (d->tile_cfg == 0 && x1 && wp) {
    const int thin = rtg_thin_kind(d);
    if (thin > 0) {
      return rtg_thin_launch(thin, d, x1, aux, wp, bias, mask, res, out, (hipStream_t)stream);
    }
  }
  if (d->tile_cfg > RTG_SCONV_CODE) {
    if (aux) return RTG_EINVAL;
    return rtg_sconv_launch(d, d->tile_cfg, x1, x2, wp, bias, mask, res, out, out2, (hipStream_t)stream);
  }
  if (d->tile_cfg > RTG_DCONV_CODE)
    return rtg_dconv_launch(d, d->tile_cfg, x1, wp, bias, mask, res, out, (hipStream_t)stream);
  if (d->tile_cfg > RTG_RESCONV_CODE && d->tile_cfg <= RTG_RESCONV_CODE + 2)
    return rtg_resconv_launch(d, d->tile_cfg - RTG_RESCONV_CODE, x1, wp, bias, mask, res, out, (hipStream_t)stream);
  ConvPlan pl;
  const int st = conv_plan(d, x1, x2, aux, wp, bias, mask, res, out, out2, &pl);
  if (st != RTG_OK) return st;
  hipStream_t s = (hipStream_t)stream;
  // a single problem is a group of one: the grouped kernel reads its arguments from the kernarg segment on demand
  // (0 SGPR spills) where a by-value ConvArgs is preloaded into SGPRs and spills 60-100 of them to VGPR lanes
  rtg_cv::GroupArgs ga;
  ga.n = 1;
  ga.p[0] = pl.a;
  for (int i = 0; i < RTG_MAX_GROUP; ++i) ga.blk_end[i] = pl.blocks;
  const size_t lds_bytes = pl.lds_bytes;
#define RTG_CASE(tm, mt, nt) \
  if (pl.TM == tm && pl.MT == mt && pl.NT == nt) return rtg_conv1d_launch_group_##tm##_##mt##_##nt(ga, lds_bytes, pl.bf, s);
  RTG_CASE(32, 1, 1) RTG_CASE(32, 1, 2) RTG_CASE(32, 1, 4) RTG_CASE(32, 2, 1) RTG_CASE(32, 2, 2)
  RTG_CASE(16, 1, 1) RTG_CASE(16, 1, 2) RTG_CASE(16, 1, 4)
#undef RTG_CASE
  return RTG_ERANGE;
}

extern "C" int rtg_conv1d_group(int n, const RtgConv1dDesc* descs, const RtgConvPtrs* ptrs, void* stream) {
  if (!descs || !ptrs) return RTG_ENULL;
  if (n < 1 || n > RTG_MAX_GROUP) return RTG_EINVAL;
  rtg_cv::GroupArgs ga;
  ga.n = n;
  size_t lds_bytes = 0;
  unsigned end = 0;
  int TM = 0, MT = 0, NT = 0, bf = 0;
  for (int i = 0; i < n; ++i) {
    if (descs[i].tile_cfg == 0) return RTG_EINVAL;          // the caller fixes ONE block shape for the whole group
    ConvPlan pl;
    const RtgConvPtrs& q = ptrs[i];
    const int st = conv_plan(&descs[i], q.x1, q.x2, q.aux, q.wp, q.bias, q.mask, q.res, q.out, q.out2, &pl);
    if (st != RTG_OK) return st;
    if (i == 0) { TM = pl.TM; MT = pl.MT; NT = pl.NT; bf = pl.bf; }
    else if (pl.TM != TM || pl.MT != MT || pl.NT != NT || pl.bf != bf) return RTG_EINVAL;
    ga.p[i] = pl.a;
    end += pl.blocks;
    ga.blk_end[i] = end;
    lds_bytes = pl.lds_bytes > lds_bytes ? pl.lds_bytes : lds_bytes;
  }
  for (int i = n; i < RTG_MAX_GROUP; ++i) ga.blk_end[i] = end;
  hipStream_t s = (hipStream_t)stream;
#define RTG_CASE(tm, mt, nt) \
  if (TM == tm && MT == mt && NT == nt) return rtg_conv1d_launch_group_##tm##_##mt##_##nt(ga, lds_bytes, bf, s);
  RTG_CASE(32, 1, 1) RTG_CASE(32, 1, 2) RTG_CASE(32, 1, 4) RTG_CASE(32, 2, 1) RTG_CASE(32, 2, 2)
  RTG_CASE(16, 1, 1) RTG_CASE(16, 1, 2) RTG_CASE(16, 1, 4)
#undef RTG_CASE
  return RTG_ERANGE;
}
