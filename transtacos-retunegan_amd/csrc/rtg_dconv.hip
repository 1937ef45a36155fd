// rtg_dconv.hip — the dense-layer implicit-GEMM Conv1d kernel ("conv kernel v2", round 3): the 128..512-channel, dilation-1
// layers at the top of the discriminators, where 75 % of their multiply-accumulates sit
//   DiscriminatorP convs.2 / .3 / .4  (128 -> 256 -> 512 -> 512, (5,1) kernels, stride 3 / 3 / 1)   discrminator.py:155-163
//   DiscriminatorS convs.5            (512 -> 512, k5, stride 1)                                     discrminator.py:44
// forward, backward-data of the stride-1 layers (same operator on RTG_PACK_DGRAD_S1 weights) and the polyphase
// backward-data of the stride-3 layers (a stride-1 operator with ceil(5/3) = 2 taps and a "shuffle" store).
//
// What the general kernel (rtg_conv1d_kernel.h) spends besides matrix instructions on these layers, and what is done here:
//   operand fetches  one 4-byte LDS read and one 4-byte weight load per v_mfma_f32_32x32x2_f32 -> ONE 16-byte fetch per
//                    FOUR matrix instructions for both operands: the 16 channels of a chunk are the four k-steps of
//                    v_mfma_f32_16x16x4_f32 (lane (kgrp, n) holds channel 4 * kq + kgrp of k-step kq), the staged patch keeps
//                    those four values of a lane adjacent, as four planes [kgrp][position][kq]: ds_read_b128 serves the
//                    lanes in groups of 16 that pair half the columns of one kgrp with the other half of the next
//                    (MI355X_MICROARCH.md, LDS), so with the planes a multiple of 256 bytes apart a group's 16 fragments
//                    are 16 consecutive positions (times the stride) = all 64 banks once (a position-major row of the four
//                    kgrp segments was a 2-way conflict on 3 of 8 lanes: half the LDS cycles), the weights come packed the same way
//                    (RtgPackJob.frag16) and are read straight from L2, one coalesced 1-KB load per 16 rows and (chunk, tap);
//   column waste     32-column tile granularity and whole-clip patches -> 16-column granularity (the tile width is chosen
//                    so that the grid is one full round of the chip: 512 x 7040 outputs are 252 tiles of 128 x 112) and a
//                    staged window of exactly the positions the tile's columns read, whatever clips it straddles;
//   geometry         run-time taps / stride / dilation / 2-D / grouping -> compile-time taps and stride, tap offsets are
//                    immediates of the LDS reads;
//   weight re-reads  waves are stacked along the output rows only, so no two waves of a block load the same weights;
//   barriers         one per 16-channel chunk, placed one tap before the chunk ends: the next chunk's patch is published
//                    while the last tap still multiplies, and the fragment prefetch never waits at a chunk boundary.
// The accumulation order of every output element — chunk, tap, channel ascending, one fused multiply-add each — is the
// general kernel's (v_mfma_f32_16x16x4_f32 chains k = 0..3 exactly like two v_mfma_f32_32x32x2_f32), and so is the
// epilogue arithmetic: results are bit-identical to every other block shape (tests/test_dconv_gpu.py).
// Exposed through rtg_conv1d as block-shape codes 8000 + 100 * shape + NT (RtgConv1dDesc.tile_cfg) when the descriptor
// says the 16-byte-fragment weight image is there (RtgConv1dDesc.wp16); the tuner times them like any other shape.
#include "rtg_common.h"
#include "rtg_dconv_kernel.h"

using namespace rtg_dc;

namespace {

// ---------------------------------------------------------------------------------------------------------------- host
bool dconv_eligible(const RtgConv1dDesc* d) {
  if (!d->wp16 || d->groups != 1 || d->C2 != 0 || d->out_split != 0 || d->tap_major) return false;
  // bf16 tensors (RtgConv1dDesc.io_bf16): with bf16 operands only; x bf16 comes activated (no pre-activation is applied to it:
  // pre_mode says what the tensor stands for, not work to do); a bf16 output is stored, not accumulated
  if (d->io_bf16 < 0 || d->io_bf16 > 15 || (d->io_bf16 != 0 && d->bf16 != 1)) return false;
  if ((d->io_bf16 & RTG_IO_OUT_BF16) && d->accumulate) return false;
  const int ckc = d->bf16 ? 32 : RTG_CK;             // channels per chunk
  const bool two_d = d->h_k > 1 || d->h_n > 1;
  if (two_d) {
    // forward of the Conv2d layers, backward-data of the row-stride-1 ones; 3 taps along the last axis
    // ... and of the row-strided ones (class-ordered clips, 2 taps of the polyphase walk along the last axis)
    if (d->h_in < 1 || d->h_k < 1 || d->h_n < 1 || d->h_pad < 0 || d->C1 % d->h_k != 0 || d->B % d->h_n != 0) return false;
    if (d->dil != 1 || d->h_stride < 1 || d->h_mode < 0 || d->h_mode > 2) return false;
    if ((d->io_bf16 & 3) != 0 && d->h_mode == 0) return false;            // (bf16 tensors: kernel-row walks only, dc_built)
    if (d->h_mode != 0 && (d->C1 / d->h_k) % ckc != 0) return false;      // whole chunks per kernel row
    if (d->h_mode == 1 && d->h_stride > 1) {
      if (d->K != 2 || d->stride != 1 || d->h_stride > 4) return false;
    } else {
      const bool k3 = d->K == 3 && (d->stride == 1 || d->stride == 2), k5 = d->K == 5 && d->stride == 3 && d->h_mode != 1;
      if (!(k3 || k5) || d->shuf_S != 1) return false;
      if (d->h_mode == 1 && d->stride != 1) return false;
    }
    if ((long long)(d->B / d->h_n) * (d->C1 / d->h_k) * d->h_in * d->L_in * 4 >= (1ll << 31)) return false;
    if ((long long)(d->B / d->h_n) * d->out_C * d->h_n * d->out_L * 4 >= (1ll << 31)) return false;
  } else {
    if (d->dil != 1 || (d->stride != 1 && d->stride != 3)) return false;
    if (!((d->K == 5) || (d->K == 2 && d->stride == 1))) return false;
    if ((long long)d->B * d->C1 * d->L_in * 4 >= (1ll << 31) || (long long)d->B * d->out_C * d->out_L * 4 >= (1ll << 31)) return false;
  }
  if (d->Cg != d->C1 || d->Cg % ckc != 0 || d->Cg < 32 || d->Mg < 64 || d->Q < min_q(d->K, two_d)) return false;
  if (d->pre_mode != RTG_PRE_NONE && d->pre_mode != RTG_PRE_LRELU) return false;
  if ((long long)d->groups * d->Mg != (long long)d->out_C * d->shuf_S) return false;
  if ((long long)d->B * d->Q >= (1ll << 30)) return false;
  return true;
}

// bf16 x: the shortest row the launch's instance is built for — 32 for the 2-D kernel-row walks (h_mode 1 / 2) on rows of at
// least 32 positions (dconv_kernel's XQ), else the shape's min_q
int xb_row_class(const RtgConv1dDesc* d, bool two_d) {
  return (two_d && d->h_mode != 0 && d->Q >= 32) ? 32 : min_q(d->K, two_d);
}

}  // namespace

#define RTG_DCONV_CODE 8000

// the block-shape codes (8000 + 100 * shape + NT16) that serve the descriptor, best guess first; returns the count
int rtg_dconv_candidates(const RtgConv1dDesc* d, int* codes, int max) {
  if (!dconv_eligible(d)) return 0;
  const long long n_cols = (long long)d->B * d->Q;
  const int n_mt16 = rtg_ceil_div(d->Mg, 16);
  struct Cand {
    int code;
    double score;
  } c[24];
  int n = 0;
  for (int si = 0; si < kNumShapes; ++si) {
    const int mb16 = kShapes[si].rw16 * kShapes[si].wb;
    if (si >= 3 && n_mt16 % mb16 != 0) continue;                     // 256-row blocks only where they divide the rows
    const int n_mb = rtg_ceil_div(n_mt16, mb16);
    for (int ni = 0; ni < 4; ++ni) {
      const int BN = kNT[ni] * 16;
      const bool two_d = d->h_k > 1 || d->h_n > 1;
      // the instances that exist (rtg_dconv_kernel.h: dc_built — the ones that do not spill, and for bf16 tensors the ones
      // the tuner ever picked)
      if (!dc_built(kShapes[si].rw16, kShapes[si].wb, kNT[ni], d->stride, two_d, d->h_mode != 0, d->bf16 != 0, d->io_bf16 & 3)) continue;
      const int pw = window_positions((int)(n_cols < BN ? n_cols : BN), d->Q, d->stride, d->K);
      const int xbt = (d->io_bf16 & RTG_IO_X_BF16) ? xb_max_tasks(BN, xb_row_class(d, two_d), d->stride, d->K, kShapes[si].wb) : 0;
      if (lds_bytes_for(pw, d->stride, kShapes[si].wb, xbt) > 158 * 1024) continue;
      const long long blocks = (long long)n_mb * ((n_cols + BN - 1) / BN);
      // rounds of the chip at one block per CU (two for the 4-wave shapes): the tail round's idle CUs are the loss
      const double slots = 256.0 * (kShapes[si].wb == 4 ? 2 : 1);
      const double rounds = (double)blocks / slots;
      const double eff = rounds / (double)(long long)(rounds + 0.999999);
      const double rows_eff = (double)n_mt16 / (double)(n_mb * mb16);
      const double cols_eff = (double)n_cols / (double)(((n_cols + BN - 1) / BN) * BN);
      c[n].code = RTG_DCONV_CODE + 100 * (si + 1) + kNT[ni];
      c[n].score = eff * rows_eff * cols_eff * (kShapes[si].rw16 >= 2 ? 1.0 : 0.95);
      ++n;
    }
  }
  int cnt = 0;
  for (int k = 0; k < n && cnt < max; ++k) {
    int bi = 0;
    for (int i = 1; i < n; ++i)
      if (c[i].score > c[bi].score) bi = i;
    if (c[bi].score < 0) break;
    codes[cnt++] = c[bi].code;
    c[bi].score = -1.0;
  }
  return cnt;
}

int rtg_dconv_launch(const RtgConv1dDesc* d, int code, const float* x, const float* wp, const float* bias,
                     const float* mask, const float* res, float* out, hipStream_t s) {
  if (!dconv_eligible(d)) return RTG_EINVAL;
  if (!x || !wp || !out) return RTG_ENULL;
  if ((reinterpret_cast<uintptr_t>(wp) & 15) != 0) return RTG_EINVAL;
  const int si = (code - RTG_DCONV_CODE) / 100 - 1, nt16 = (code - RTG_DCONV_CODE) % 100;
  if (si < 0 || si >= kNumShapes) return RTG_EINVAL;
  const int rw16 = kShapes[si].rw16, wb = kShapes[si].wb, BN = nt16 * 16;
  DArgs a;
  // the 16-byte-fragment image follows the standard image of the layer (RtgConv1dDesc.wp16)
  const long long std_size = d->bf16 ? rtg_packed_size_bf16(1, d->Mg, d->Cg, d->K, d->tile_m)
                                     : rtg_packed_size(1, d->Mg, d->Cg, d->K, d->tile_m);
  if (std_size < 0 || (std_size & 3) != 0) return RTG_EINVAL;
  a.x = x; a.wp = wp + std_size; a.bias = bias; a.mask = mask; a.res = res; a.out = out;
  const bool xb = (d->io_bf16 & RTG_IO_X_BF16) != 0, ob = (d->io_bf16 & RTG_IO_OUT_BF16) != 0;
  a.mask_b16 = (d->io_bf16 & RTG_IO_MASK_BF16) ? 1 : 0; a.res_b16 = (d->io_bf16 & RTG_IO_RES_BF16) ? 1 : 0;
  a.enc_slope = d->enc_slope;
  const bool two_d = d->h_k > 1 || d->h_n > 1;
  a.h_in = two_d ? d->h_in : 1; a.h_k = two_d ? d->h_k : 1; a.h_stride = two_d ? d->h_stride : 1;
  a.h_pad = two_d ? d->h_pad : 0; a.h_n = two_d ? d->h_n : 1; a.h_mode = two_d ? d->h_mode : 0;
  a.n_co = d->C1 / a.h_k;                            // real channels (backward-data: output channels of the layer)
  const int ckc = d->bf16 ? 32 : RTG_CK;
  a.cpk = a.n_co / ckc;
  for (int c = 0, base = 0; c < 4; ++c) {
    int f = (c - a.h_pad) % a.h_stride;
    if (f < 0) f += a.h_stride;
    const int n = (c < a.h_stride && f < a.h_n) ? (a.h_n - f + a.h_stride - 1) / a.h_stride : 0;
    a.cls_f[c] = f; a.cls_n[c] = n > 0 ? n : 1; a.cls_base[c] = base;
    base += (d->B / a.h_n) * n;
  }
  a.B = d->B; a.C = d->C1 / a.h_k; a.L_in = d->L_in; a.Mg = d->Mg; a.n_cc = d->Cg / ckc; a.Q = d->Q; a.pad = d->pad;
  a.out_C = d->out_C; a.out_L = d->out_L; a.shuf_S = d->shuf_S; a.shuf_P = d->shuf_P;
  a.pre = d->pre_mode == RTG_PRE_LRELU ? 1 : 0; a.act = d->act; a.accumulate = d->accumulate;
  a.pre_slope = d->pre_slope; a.mask_slope = d->mask_slope; a.out_scale = d->out_scale; a.act_slope = d->act_slope;
  a.seg_pw = (d->Q - 1) * d->stride + d->K;
  a.n_cols = d->B * d->Q;
  const int n_mt16 = rtg_ceil_div(d->Mg, 16);
  a.n_mb = rtg_ceil_div(n_mt16, rw16 * wb);
  const long long total = (long long)a.n_mb * rtg_ceil_div(a.n_cols, BN);
  if (total > (1ll << 28)) return RTG_ERANGE;
  a.total = (int)total;
  a.PW = window_positions(a.n_cols < BN ? a.n_cols : BN, d->Q, d->stride, d->K);
  // (bf16 x: + the 16 readable bytes the caller guarantees behind the tensor — a 16-byte load that starts at its last elements
  // runs past the end, and the range check drops whole dwords: clipped at the end it would lose the last element)
  a.x_bytes = (d->B / a.h_n) * a.C * a.h_in * d->L_in * (xb ? 2 : 4) + (xb ? 16 : 0);       // 1-D: h_n = h_in = 1
  const int out_elems = d->B * d->out_C * d->out_L;               // (B = items * h_n)
  a.out_bytes = out_elems * (ob ? 2 : 4);
  a.mask_bytes = out_elems * (a.mask_b16 ? 2 : 4); a.res_bytes = out_elems * (a.res_b16 ? 2 : 4);
  // bf16 input: the positions 0 .. xb_Lv - 1 of a row are the ones a clip's segment reads, in units of 8
  a.xb_Lv = a.seg_pw - d->pad < d->L_in ? a.seg_pw - d->pad : d->L_in;
  if (a.xb_Lv < 1) a.xb_Lv = 1;
  a.xb_ups = (a.xb_Lv + 7) / 8;
  a.xb_j0 = a.xb_n0 = a.xb_units = 0;                             // (per block: computed in the kernel)
  a.xq = (xb && xb_row_class(d, two_d) == 32) ? 32 : 0;
  const size_t lds_bytes = lds_bytes_for(a.PW, d->stride, wb, xb ? xb_max_tasks(BN, xb_row_class(d, two_d), d->stride, d->K, wb) : 0);
  if (lds_bytes > 158 * 1024) return RTG_ERANGE;
  // ---- the XCDs' item ranges: equal work.  Items are (column tile, row block) with the row block fastest; a column tile's
  // work is the chunks it walks: all of them, or — class-ordered clips, the tile inside one residue class — that class's
  // kernel rows.  The classes are contiguous column ranges, so the cumulative work is piecewise linear: a few segments.
  {
    const int n_tiles = (int)(total / a.n_mb);
    struct Seg { int tiles; long long work; } seg[16];
    int ns = 0;
    const bool cls = two_d && a.h_mode == 1 && a.h_stride > 1;
    // work of a tile in matrix-pipe cycles of one wave: its chunks' matrix instructions plus a fixed part (prologue, epilogue:
    // memory time, which does not scale with the kernel rows — the bf16 instances and the thin first layers are mostly that)
    const long long chunk_cyc = (long long)rw16 * nt16 * d->K * (d->bf16 ? 16 : 4 * 32);
    // (measured, same box: fixed part 0 / 6000 / 20000 / equal counts: config 4 56.97 / 56.73 / 56.89 / 58.13 ms, config 3
    // 48.82 / 47.57 / 47.72 / 48.14)
    const long long fixed_cyc = 6000;
    auto tile_work = [&](int chunks) { return chunks * chunk_cyc + fixed_cyc; };
    if (!cls) {
      seg[ns++] = {n_tiles, 1};
    } else {
      int t = 0;                                                   // tiles accounted for
      for (int c = 0; c < a.h_stride && c < 4; ++c) {
        const long long lo = (long long)a.cls_base[c] * a.Q;
        const long long hi = c + 1 < a.h_stride && c + 1 < 4 ? (long long)a.cls_base[c + 1] * a.Q : (long long)a.n_cols;
        if (hi <= lo) continue;
        const int t_lo = (int)((lo + BN - 1) / BN);               // first tile that starts inside the class
        int t_hi = (int)(hi / BN);                                 // tiles [t_lo, t_hi) end inside it
        if (hi == (long long)a.n_cols) t_hi = n_tiles;             // (the partial last tile)
        if (t_lo > t) { seg[ns++] = {t_lo - t, tile_work(a.n_cc)}; t = t_lo; }   // straddling tiles before it: every chunk
        if (t_hi > t) {
          const int rows = c < a.h_k ? (a.h_k - c + a.h_stride - 1) / a.h_stride : 0;
          seg[ns++] = {t_hi - t, tile_work(rows * a.cpk)};
          t = t_hi;
        }
      }
      if (t < n_tiles) seg[ns++] = {n_tiles - t, tile_work(a.n_cc)};
    }
    long long work = 0;
    for (int i = 0; i < ns; ++i) work += (long long)seg[i].tiles * seg[i].work;
    a.xcd_first[0] = 0;
    int si2 = 0, t_done = 0;                                       // segment cursor: tiles before it, work before it
    long long w_done = 0;
    for (int x = 1; x < 8; ++x) {
      const long long target = (work * x + 7) / 8;
      while (si2 < ns && w_done + (long long)seg[si2].tiles * seg[si2].work < target) {
        w_done += (long long)seg[si2].tiles * seg[si2].work;
        t_done += seg[si2].tiles;
        ++si2;
      }
      int tile = n_tiles;
      if (si2 < ns) tile = t_done + (int)((target - w_done + seg[si2].work - 1) / seg[si2].work);
      if (tile > n_tiles) tile = n_tiles;
      a.xcd_first[x] = tile * a.n_mb;
    }
    a.xcd_first[8] = a.total;
  }
  int per_max = 0;
  for (int x = 0; x < 8; ++x) per_max = a.xcd_first[x + 1] - a.xcd_first[x] > per_max ? a.xcd_first[x + 1] - a.xcd_first[x] : per_max;
  const unsigned blocks = (unsigned)(8 * per_max);
  if (d->io_bf16 & 3) {
    const int io = d->io_bf16 & 3;
    return io == 1 ? rtg_dconv_launch_io1(a, si, nt16, d->stride, d->K, two_d, blocks, lds_bytes, s)
         : io == 2 ? rtg_dconv_launch_io2(a, si, nt16, d->stride, d->K, two_d, blocks, lds_bytes, s)
                   : rtg_dconv_launch_io3(a, si, nt16, d->stride, d->K, two_d, blocks, lds_bytes, s);
  }
  return d->bf16 ? rtg_dconv_launch_bf(a, si, nt16, d->stride, d->K, two_d, blocks, lds_bytes, s)
                 : launch_shape<false, 0>(a, si, nt16, d->stride, d->K, two_d, blocks, lds_bytes, s);
}

extern "C" long long rtg_packed_size_frag16(int Mg, int Cg, int K) {
  if (Mg < 1 || Cg < 1 || K < 1) return RTG_EINVAL;
  return (long long)((Mg + 15) / 16) * ((Cg + RTG_CK - 1) / RTG_CK) * K * 256;
}

// bf16 fragments: 1 KB per (16-row tile, 32-channel chunk, tap), in floats
extern "C" long long rtg_packed_size_frag16_bf16(int Mg, int Cg, int K) {
  if (Mg < 1 || Cg < 1 || K < 1) return RTG_EINVAL;
  return (long long)((Mg + 15) / 16) * ((Cg + 31) / 32) * K * 256;
}
