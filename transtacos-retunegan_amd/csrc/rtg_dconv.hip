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

namespace {

using rsrc_t = __amdgpu_buffer_rsrc_t;
#define DC_OOB 0x80000000u
// floats between the four kgrp planes of a patch buffer: the positions' 16-byte fragments, rounded up to 256 bytes (odd
// strides: a lane group's fragments n * S * 16 bytes fill the banks exactly) plus 16 bytes for the even stride (the two
// kgrp halves of a group then take the even and the odd 16-byte bank quads)
constexpr int plane_floats(int PW, int S) { return ((PW * 4 + 63) / 64) * 64 + ((S & 1) ? 0 : 4); }

struct DArgs {
  const float *x, *wp, *bias, *mask, *res;
  float* out;
  int B, C, L_in, Mg, n_cc, Q, pad, out_C, out_L, shuf_S, shuf_P;
  int pre, act, accumulate;
  float pre_slope, mask_slope, out_scale, act_slope;
  int seg_pw;                 // virtual positions per clip: (Q - 1) * S + K
  int n_cols;                 // B * Q
  int n_mb, total, per_xcd;   // row blocks, work items, work items per XCD
  int PW;                     // staged positions per buffer
  int x_bytes, out_bytes;
  // second dimension (RtgConv1dDesc.h_*): a clip is an (item, output row) pair, a channel a (channel, kernel row) pair
  int h_in, h_k, h_stride, h_pad, h_n, h_mode, n_co;
  // class-ordered clips (backward-data over a row-strided layer): output row r only receives kernel rows kh == (r + h_pad)
  // (mod h_stride), so the clip sequence lists the rows of residue class 0 of every item first, then class 1, ...: a
  // column tile inside one class walks only that class's kernel rows.  Class c: first row cls_f, cls_n rows per item,
  // clips [cls_base, ...); cpk = 16-channel chunks per kernel row
  int cls_f[4], cls_n[4], cls_base[4], cpk;
};

__device__ __forceinline__ float dc_load(rsrc_t r, unsigned off, unsigned soff = 0) {
#if defined(RTG_EXP_DC_LINEAR)           // ablation: the address math dropped, a coalesced in-range load instead
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (threadIdx.x & 63u) * 4u, 0, 0));
#elif defined(RTG_EXP_DC_KEEPMATH)       // ablation: the address math kept alive, the load coalesced
  asm volatile("" ::"v"(off), "s"(soff));
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (threadIdx.x & 63u) * 4u, 0, 0));
#else
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, soff, 0));
#endif
}

// shortest row served (bounds the clip boundaries a column tile can straddle, hence the staging registers): 8 for the k5 /
// 2-tap 1-D layers, 4 for the 3-tap rows of the spectrogram discriminators (5 columns after their strided layers)
constexpr int min_q(int K, bool two_d = false) { return (K == 3 || two_d) ? 4 : 8; }

// positions a column tile of `cols` columns reads: the span of its columns' virtual positions plus the taps; every clip
// boundary inside the tile adds the gap between two clips' segments (seg_pw - Q * S = K - S)
constexpr int window_positions(int cols, int Q, int S, int K) {
  const int crossings = Q >= cols ? 1 : (cols - 2) / Q + 1;          // most clip boundaries between the first and last column
  return (cols - 1) * S + K + crossings * (K - S);
}

// RW16: 16-row tiles per wave; WB: waves per block (stacked along the rows); NT16: 16-column tiles per block (= per wave);
// S: stride of the B-operand walk; K: taps; TWO_D: the Conv2d layers of StftDiscriminator run along their last axis
// (discrminator.py:255-262), the patch row of clip (item, r) and channel (c, kh) being input row r * h_stride - h_pad + kh
// (forward) or r + h_pad - kh (backward-data of a row-stride-1 layer, channels ordered (kh, c))
//
// BF (RtgConv1dDesc.bf16, BASELINE configs[2]): bf16 operands on v_mfma_f32_16x16x32_bf16, fp32 accumulation.  A chunk is 32
// channels: lane (kgrp, n) holds channels 8 * kgrp .. + 7 of the chunk as ONE 16-byte fragment, so the patch planes, the
// fragment reads, the weight loads (image [16-row tile][32-channel chunk][tap][kgrp][row][8 bf16]) and the loop are the fp32
// kernel's with one matrix instruction per fragment pair instead of four; a wave stages 8 channels per position (fp32
// tensors in HBM, activation applied in fp32, rounded to nearest even when the 16 bytes are written to LDS).
using bf16x8 = __bf16 __attribute__((ext_vector_type(8)));
// HB: 2-D backward-data (RtgConv1dDesc.h_mode 1) — a template parameter although it only selects address arithmetic: with
// both forms in one loop the compiler's wait-count bookkeeping merged their pending loads at every join and waited for the
// staging loads (and the fragment loads behind them) a chunk early
template <int RW16, int WB, int NT16, int S, int K, bool TWO_D, bool CLS, bool BF, bool HB>
__global__ __launch_bounds__(WB * 64, 2) void dconv_kernel(const DArgs a) {
  static_assert(!CLS || TWO_D, "class-ordered clips belong to the 2-D backward-data");
  static_assert((!CLS || HB) && (!HB || TWO_D), "h_mode 1 is 2-D; class-ordered clips are backward-data");
  constexpr int CKC = BF ? 32 : RTG_CK;              // channels per chunk
  constexpr int NSI = BF ? 8 : 4;                    // channels a wave stages per position
  constexpr int BN = NT16 * 16;
  // positions staged per lane: enough for the widest window of the shape (rows of min_q(K, TWO_D) positions); iterations past
  // the actual window load nothing (out-of-range offsets) and write nothing
  constexpr int MAXIT = (window_positions(BN, min_q(K, TWO_D), S, K) + 64 * (WB / 4) - 1) / (64 * (WB / 4));
  constexpr int SPI = 64 * (WB / 4);                 // positions staged per iteration by the WB / 4 waves of a channel group
  constexpr int TW = K >= 3 ? K - 2 : 0;             // tap after which the next chunk's patch is written and published
  extern __shared__ __attribute__((aligned(16))) float lds[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // block -> work item: blocks b and b + 8 share an XCD, each XCD walks a contiguous range of items, the row blocks of
  // one column tile next to each other (they read the same input window: L2 hits)
  const int item = (int)(blockIdx.x & 7u) * a.per_xcd + (int)(blockIdx.x >> 3);
  if (item >= a.total) return;
  const int mb = item % a.n_mb, nt = item / a.n_mb;
  const int n0 = nt * BN;
  const int clip0 = n0 / a.Q, q0 = n0 - clip0 * a.Q;
  const int g0 = q0 * S;                             // virtual position (within clip0's segment) of LDS position 0
  const int planeF = plane_floats(a.PW, S);
  const int bufF = 4 * planeF;                       // floats per LDS buffer

  // ---- staging geometry: LDS position o <-> (clip, input position); a wave stages channels kgrp, kgrp + 4, + 8, + 12 of
  // the chunk (one 16-byte LDS row segment per position)
  const int skgrp = wave & 3;
  // clip of the (possibly class-ordered) sequence -> (item, row of the output tensor, residue class)
  auto decode = [&](int cl, int& item, int& r, int& cls) __attribute__((always_inline)) {
    cls = 0;
    if constexpr (CLS) {
#pragma unroll
      for (int c = 1; c < 4; ++c)
        if (c < a.h_stride && cl >= a.cls_base[c]) cls = c;
      const int idx = cl - a.cls_base[cls];
      item = idx / a.cls_n[cls];
      r = a.cls_f[cls] + (idx - item * a.cls_n[cls]) * a.h_stride;
    } else {
      item = cl / a.h_n;
      r = cl - item * a.h_n;
    }
  };
  unsigned soff[MAXIT];                              // byte offset of (clip, channel 0, position) in x, or out of range
  int srow[TWO_D ? MAXIT : 1];                       // 2-D: the input row kernel row 0 reads for this position's clip
  int scls[CLS ? MAXIT : 1];                         // class-ordered: the residue class of this position's row
#pragma unroll
  for (int it = 0; it < MAXIT; ++it) {
    const int o = (wave >> 2) * 64 + lane + SPI * it;
    const int G = g0 + o;
    const int seg = G / a.seg_pw, w = G - seg * a.seg_pw;
    const int clip = clip0 + seg, pos = w - a.pad;
    const bool ok = o < a.PW && clip < a.B && pos >= 0 && pos < a.L_in;
    if constexpr (TWO_D) {
      int item, ho, cls;
      decode(clip, item, ho, cls);
      srow[it] = !HB ? ho * a.h_stride - a.h_pad : ho + a.h_pad;
      if constexpr (CLS) {
        // rows of class cls take kernel rows cls, cls + h_stride, ...: kernel row cls + m * h_stride reads row srow - m
        srow[it] = (ho + a.h_pad - cls) / a.h_stride;
        scls[it] = cls;
      }
      // (item, channel 0, row srow, position) — wrapping arithmetic, the row becomes valid once the kernel row is added;
      // the channel's rows are a wave-uniform offset of the load.  An invalid position has no valid row.
      soff[it] = ((unsigned)item * (unsigned)a.C * (unsigned)a.h_in * (unsigned)a.L_in + (unsigned)pos) * 4u +
                 (unsigned)srow[it] * (unsigned)a.L_in * 4u;
      if (!ok) srow[it] = -(1 << 28);
    } else {
      soff[it] = ok ? ((unsigned)clip * (unsigned)a.C * (unsigned)a.L_in + (unsigned)pos) * 4u : DC_OOB;
    }
  }
  const rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
  const unsigned chb = (unsigned)a.L_in * 4u;        // bytes per channel row
  // staging registers: one set (the patch of chunk v + 2 is requested during the last tap of chunk v and written during
  // chunk v + 1), or two alternating sets (BF: a chunk's matrix instructions last a few hundred cycles, less than the
  // latency of the loads: chunk v + 3 is requested at the end of chunk v)
#ifndef RTG_DC_NSET_MODE
#define RTG_DC_NSET_MODE 0
#endif
  constexpr int NSET = (BF || (RTG_DC_NSET_MODE == 1 && K == 2) || (RTG_DC_NSET_MODE == 2 && (K == 2 || S == 3)) || RTG_DC_NSET_MODE == 3) ? 2 : 1;
  float st[NSET][NSI][MAXIT];
  // (unconditional: past the last chunk the loads go out of range and return zeros that nobody writes — a branch around
  // them would make the compiler's vmcnt bookkeeping pessimistic for every weight fetch after the join)
  // 2-D: which (channel, kernel row) a staged virtual channel is.  Backward-data orders them (kernel row, channel) with whole
  // chunks per kernel row: the kernel row is the walk's (Walk::kh, uniform over the chunk).  Forward orders them (channel,
  // kernel row): sub-channel i of this wave starts at virtual channel base_i and moves CKC channels per chunk — kept as a
  // (channel, kernel row) pair advanced chunk by chunk (every division here was ~40 vector instructions per staged channel
  // and chunk: four times the bf16 kernel's matrix time)
  struct Walk {
    int rc;                   // real chunk (index into the weight image); n_cc once past the end
    int kh, khq, khr, cw;     // backward-data: kernel row, kh / h_stride, kh % h_stride, chunk within the kernel row
  };
  [[maybe_unused]] int m0c[NSI], m0r[NSI], m0q = 0, m0rem = 0;
  if constexpr (TWO_D && !HB) {
    {
      m0q = CKC / a.h_k;
      m0rem = CKC - m0q * a.h_k;
#pragma unroll
      for (int i = 0; i < NSI; ++i) {
        const int vc0 = BF ? 8 * skgrp + i : skgrp + 4 * i;
        m0c[i] = vc0 / a.h_k;
        m0r[i] = vc0 - m0c[i] * a.h_k;
      }
    }
  }
  // (unconditional: past the last chunk the loads go out of range and return zeros that nobody writes — a branch around
  // them would make the compiler's vmcnt bookkeeping pessimistic for every weight fetch after the join)
  auto stage_issue = [&](const Walk& w, auto set_tag) __attribute__((always_inline)) {
    constexpr int SET = decltype(set_tag)::value;
#ifdef RTG_EXP_DC_NOSTAGE
    return;
#endif
    const bool is_past = w.rc >= a.n_cc;
    if constexpr (TWO_D) {
      if constexpr (!HB) {
        {
#pragma unroll
          for (int i = 0; i < NSI; ++i) {
            const int kh = is_past ? (1 << 24) : m0r[i];
            const unsigned khb = (unsigned)kh * chb, cb = (unsigned)(m0c[i] * a.h_in) * chb;
#pragma unroll
            for (int it = 0; it < MAXIT; ++it) {
              const bool ok = (unsigned)(srow[it] + kh) < (unsigned)a.h_in;
              st[SET][i][it] = dc_load(rx, ok ? soff[it] + khb : DC_OOB, is_past ? 0u : cb);
            }
            const int r2 = m0r[i] + m0rem;
            const bool wrap = r2 >= a.h_k;
            m0r[i] = wrap ? r2 - a.h_k : r2;
            m0c[i] += wrap ? m0q + 1 : m0q;
          }
        }
      } else {
        // one kernel row per chunk: the rows (and, class-ordered, whether the row's class takes this kernel row) once
        const int dr = is_past ? (1 << 24) : (CLS ? -w.khq : -w.kh);
        unsigned voff[MAXIT];
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
          bool ok = (unsigned)(srow[it] + dr) < (unsigned)a.h_in;
          if constexpr (CLS) ok = ok && scls[it] == w.khr;       // a kernel row of another residue class: zeros
          voff[it] = ok ? soff[it] + (unsigned)dr * chb : DC_OOB;
        }
#pragma unroll
        for (int i = 0; i < NSI; ++i) {
          const int c = w.cw * CKC + (BF ? 8 * skgrp + i : skgrp + 4 * i);
          const unsigned cb = is_past ? 0u : (unsigned)(c * a.h_in) * chb;
#pragma unroll
          for (int it = 0; it < MAXIT; ++it) st[SET][i][it] = dc_load(rx, voff[it], cb);
        }
      }
    } else {
      const unsigned past = is_past ? DC_OOB : 0u;
#pragma unroll
      for (int i = 0; i < NSI; ++i) {
        const int vc = w.rc * CKC + (BF ? 8 * skgrp + i : skgrp + 4 * i);
        const unsigned coff = (unsigned)vc * chb | past;
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) st[SET][i][it] = dc_load(rx, (soff[it] + coff) | (soff[it] & DC_OOB));
      }
    }
  };
  const float wslope = a.pre ? a.pre_slope : 1.f;
  auto stage_write = [&](float* buf, auto set_tag) __attribute__((always_inline)) {
    constexpr int SET = decltype(set_tag)::value;
#ifdef RTG_EXP_DC_NOSTWRITE
    return;
#endif
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
      const int o = (wave >> 2) * 64 + lane + SPI * it;
      if (o < a.PW) {
        f32x4 v;
        if constexpr (BF) {
          bf16x8 h;
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            float t = st[SET][i][it];
            asm volatile("" : "+v"(t));
            h[i] = (__bf16)(t > 0.f ? t : t * wslope);
          }
          v = __builtin_bit_cast(f32x4, h);
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float t = st[SET][i][it];
            asm volatile("" : "+v"(t));                // keep the consumption (and its wait) here, below the multiplications
            v[i] = t > 0.f ? t : t * wslope;
          }
        }
        *reinterpret_cast<f32x4*>(buf + skgrp * planeF + o * 4) = v;
      }
    }
  };

  // ---- operand addressing
  const int n16 = lane & 15, kgrp = lane >> 4;
  int bcol[NT16];                                    // float offset of this lane's fragment of column tile j at tap 0
#pragma unroll
  for (int j = 0; j < NT16; ++j) {
    int n = n0 + j * 16 + n16;
    if (n > a.n_cols - 1) n = a.n_cols - 1;          // junk column: a valid position, dropped in the epilogue
    const int clip = n / a.Q, q = n - clip * a.Q;
    bcol[j] = ((clip - clip0) * a.seg_pw + q * S - g0) * 4 + kgrp * planeF;
  }
  const int n_mt16 = (a.Mg + 15) >> 4;
  const f32x4* aptr[RW16];
#pragma unroll
  for (int i = 0; i < RW16; ++i) {
    int mt = (mb * WB + wave) * RW16 + i;
    if (mt > n_mt16 - 1) mt = n_mt16 - 1;            // clamped duplicate tile, dropped in the epilogue
    aptr[i] = reinterpret_cast<const f32x4*>(a.wp) + (size_t)mt * a.n_cc * K * 64 + lane;
  }
  // the chunks this block walks: all of them, or (class-ordered clips, every column of the tile in ONE residue class) only
  // the kernel rows of that class — channels are ordered (kernel row, channel), so those are whole chunk ranges
  int n_v = a.n_cc;
  [[maybe_unused]] int cls_blk = 0;
  [[maybe_unused]] bool pure = false;
  if constexpr (CLS) {
    int it0, r0, c0, it1, r1, c1;
    const int n_last = (n0 + BN < a.n_cols ? n0 + BN : a.n_cols) - 1;
    decode(clip0, it0, r0, c0);
    decode(n_last / a.Q, it1, r1, c1);
    pure = c0 == c1;
    cls_blk = c0;
    if (pure) n_v = (c0 < a.h_k ? (a.h_k - c0 + a.h_stride - 1) / a.h_stride : 0) * a.cpk;
  }
  const int n_vp = ((K & 1) || NSET == 2) ? (n_v + 1) & ~1 : n_v;   // chunks the loop walks (an even count where it is unrolled by two)
  // generator of the walk: virtual chunk 0, 1, 2, ... -> real chunk and (2-D backward-data) its kernel row, kept as
  // counters (no division per chunk)
  int gv = 0;
  [[maybe_unused]] int gk = 0, gw = 0, gq = 0, gr = 0;
  auto gen = [&]() __attribute__((always_inline)) {
    Walk w{a.n_cc, 0, 0, 0, 0};
    const bool live = gv < n_v;
    if constexpr (HB) {
      bool p = false;
      if constexpr (CLS) p = pure;
      w.cw = gw;
      w.kh = p ? cls_blk + gk * a.h_stride : gk;
      w.khq = p ? gk : gq;
      w.khr = p ? cls_blk : gr;
      w.rc = live ? w.kh * a.cpk + gw : a.n_cc;
      const bool wrap_w = gw + 1 == a.cpk;
      const bool wrap_r = wrap_w && gr + 1 == a.h_stride;
      gw = wrap_w ? 0 : gw + 1;
      gk += wrap_w ? 1 : 0;
      gr = wrap_r ? 0 : gr + (wrap_w ? 1 : 0);
      gq += wrap_r ? 1 : 0;
    } else {
      w.rc = live ? gv : a.n_cc;
    }
    ++gv;
    return w;
  };

  f32x4 acc[RW16][NT16];
#pragma unroll
  for (int i = 0; i < RW16; ++i)
#pragma unroll
    for (int j = 0; j < NT16; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  struct Frag {
    f32x4 a[RW16], b[NT16];
  };
  // the fragments of (chunk rc, tap t) — step s = rc * K + t of the weight image: RW16 coalesced 1-KB weight loads from
  // L2, NT16 16-byte LDS reads
  const int n_steps = a.n_cc * K;
  auto fetch = [&](Frag& f, int s, const float* bsrc) __attribute__((always_inline)) {
    const int sc = s < n_steps ? s : n_steps - 1;    // (past the end: re-read the last step, never used)
#ifndef RTG_EXP_DC_NOA
#pragma unroll
    for (int i = 0; i < RW16; ++i) f.a[i] = aptr[i][(size_t)sc * 64];
#endif
#ifndef RTG_EXP_DC_NOB
#pragma unroll
    for (int j = 0; j < NT16; ++j) f.b[j] = *reinterpret_cast<const f32x4*>(bsrc + bcol[j]);
#endif
  };
  auto mma = [&](const Frag& f) __attribute__((always_inline)) {
#ifdef RTG_EXP_DC_NOMMA
    return;
#endif
    // (inline asm with the accumulator tied to the destination: left to itself the register allocator lets the bf16 form —
    // and the strided fp32 instances — write a product into the registers of a dead fragment, copies every accumulator and
    // fragment back at the loop's back edge and waits for ALL loads there, the staged patch two chunks ahead included)
    if constexpr (BF) {
#pragma unroll
      for (int i = 0; i < RW16; ++i)
#pragma unroll
        for (int j = 0; j < NT16; ++j)
          asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i][j]) : "v"(f.a[i]), "v"(f.b[j]));
    } else {
#pragma unroll
      for (int kq = 0; kq < 4; ++kq)
#pragma unroll
        for (int i = 0; i < RW16; ++i)
#pragma unroll
          for (int j = 0; j < NT16; ++j) {
            const float av = f.a[i][kq], bv = f.b[j][kq];
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i][j]) : "v"(av), "v"(bv));
          }
    }
  };

  // ---- prologue: chunk 0 staged and published, chunk 1 requested, fragments of step 0 fetched
  using Set0 = std::integral_constant<int, 0>;
  using Set1 = std::integral_constant<int, NSET - 1>;
  Walk rc0 = gen(), rc1 = gen(), rc2 = gen();         // the current virtual chunk, the next, the one after
  [[maybe_unused]] Walk rc3 = rc2;                    // (two sets: and the one after that)
  if constexpr (NSET == 2) rc3 = gen();
  stage_issue(rc0, Set0{});
  stage_write(lds, Set0{});
  __syncthreads();
  stage_issue(rc1, Set1{});                           // (one set: into the registers just written out)
  if constexpr (NSET == 2) stage_issue(rc2, Set0{});
  Frag f0, f1;
#if defined(RTG_EXP_DC_NOA) || defined(RTG_EXP_DC_NOB)
  for (int i = 0; i < RW16; ++i) f0.a[i] = f1.a[i] = f32x4{1.f, 1.f, 1.f, 1.f};
  for (int j = 0; j < NT16; ++j) f0.b[j] = f1.b[j] = f32x4{1.f, 1.f, 1.f, 1.f};
#endif
  fetch(f0, rc0.rc * K, lds);

  // one chunk (virtual index v): K taps; `cur` holds the fragments of tap 0 on entry, and of the next chunk's tap 0 on
  // exit (in `cur` again when K is even, in `oth` when K is odd: the caller alternates)
  // `nset`: the register set that holds the next chunk's patch (and takes the request issued at the end of this chunk)
  auto chunk = [&](int v, Frag& cur, Frag& oth, auto nset) __attribute__((always_inline)) {
    const float* bufc = lds + (v & 1) * bufF;
    float* bufn = lds + ((v + 1) & 1) * bufF;
#pragma unroll
    for (int t = 0; t < K; ++t) {
      Frag& fc = (t & 1) ? oth : cur;
      Frag& fn = (t & 1) ? cur : oth;
      // request the next step's fragments, THEN (last tap) the patch of the chunk after the next: the wait for the
      // fragments one step later does not include the patch loads (vmcnt retires in order)
      if (t + 1 < K) fetch(fn, rc0.rc * K + t + 1, bufc + (t + 1) * 4);
      else fetch(fn, rc1.rc * K, bufn);
      if (t == K - 1) stage_issue(NSET == 2 ? rc3 : rc2, nset);
      __builtin_amdgcn_sched_barrier(0);
      mma(fc);
      __builtin_amdgcn_sched_barrier(0);
      if (t == TW) {
        // publish the next chunk's patch: its buffer was last read by fragment fetches that completed before the
        // previous chunk's barrier; the reads of this chunk's last tap (just requested) are waited for here too
        if (v + 1 < n_vp) stage_write(bufn, nset);
        // (one asm statement: nothing can be scheduled between the wait and the barrier, no memory access across it)
#ifndef RTG_EXP_DC_NOBAR
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if constexpr (NSET == 2) { rc0 = rc1; rc1 = rc2; rc2 = rc3; rc3 = gen(); }
    else { rc0 = rc1; rc1 = rc2; rc2 = gen(); }
  };
  // chunk v + 1's patch sits in set (v + 1) % NSET: odd chunks in Set1, even ones in Set0.  The loop body is two chunks
  // (the fragment sets swap with an odd tap count, the staging sets alternate); an odd walk gets one chunk past the end —
  // out-of-range loads, a patch of zeros — instead of a tail copy of the body: with a tail the register allocator copied
  // all accumulators and fragments at the loop header and waited for every load in flight there
  int cc = 0;
  if constexpr ((K & 1) || NSET == 2) {
    do {                                    // (n_vp >= 2)
      chunk(cc, f0, f1, Set1{});
      if constexpr (K & 1) chunk(cc + 1, f1, f0, Set0{});
      else chunk(cc + 1, f0, f1, Set0{});
      cc += 2;
    } while (cc < n_vp);
  } else {
    for (; cc < n_v; ++cc) chunk(cc, f0, f1, Set0{});
  }

  // (the matrix instructions are inline asm: the compiler does not know their results are still in flight)
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#ifdef RTG_EXP_DC_NOEPI                                 // ablation: everything but the epilogue (results are not stored)
  if (a.B > 0) return;
#endif
  // ---- epilogue: out = act(((acc + bias) * dmask + res) * out_scale) (+ out), the arithmetic and rounding of the general
  // kernel; 32-bit element offsets through buffer descriptors, invalid rows / columns go to an out-of-range offset the
  // hardware drops.  Row m' of the GEMM is output channel m' / S_out at phase m' % S_out (polyphase backward-data).
  const rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, a.out_bytes, 0x00020000);
  const rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)(a.bias ? a.bias : a.out), 0, a.bias ? a.out_C * 4 : 0, 0x00020000);
  const rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc((void*)(a.mask ? a.mask : a.out), 0, a.mask ? a.out_bytes : 0, 0x00020000);
  const rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void*)(a.res ? a.res : a.out), 0, a.res ? a.out_bytes : 0, 0x00020000);
  const float mslope = a.mask ? a.mask_slope : 1.f;
  const int So = a.shuf_S;
  const float invS = 1.0f / (float)So;
#pragma unroll
  for (int i = 0; i < RW16; ++i) {
    const int mt = (mb * WB + wave) * RW16 + i;
    if (mt >= n_mt16) continue;
    float bv[4];
    unsigned rowoff[4];
    int rowph[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = mt * 16 + kgrp * 4 + r;
      int ch = m, ph = 0;
      if (So != 1) {                                  // m / S through the float reciprocal (m < 2^24), one correction step
        ch = (int)((float)m * invS);
        ph = m - ch * So;
        if (ph < 0) { --ch; ph += So; }
        else if (ph >= So) { ++ch; ph -= So; }
        ph -= a.shuf_P;
      }
      const bool rok = m < a.Mg;
      rowoff[r] = (unsigned)(ch * a.h_n * a.out_L + ph) * 4u;          // (h_n == 1 in 1-D)
      rowph[r] = rok ? ph : -(1 << 28);
      bv[r] = dc_load(rb, rok ? (unsigned)ch * 4u : DC_OOB);
    }
#pragma unroll
    for (int j = 0; j < NT16; ++j) {
      const int n = n0 + j * 16 + n16;
      const int clip = n / a.Q, q = n - clip * a.Q;
      const int qs = n < a.n_cols ? q * So : -(1 << 28);
      unsigned col;
      if constexpr (TWO_D) {                               // clip -> (item, output row) of [items, out_C, h_n, out_L]
        int item, ho, cls;
        decode(clip, item, ho, cls);
        col = ((unsigned)(item * a.out_C * a.h_n + ho) * (unsigned)a.out_L + (unsigned)(q * So)) * 4u;
      } else {
        col = ((unsigned)(clip * a.out_C) * (unsigned)a.out_L + (unsigned)(q * So)) * 4u;
      }
      unsigned off[4];
      float mv[4], rv[4], av[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) off[r] = ((unsigned)(qs + rowph[r]) < (unsigned)a.out_L) ? col + rowoff[r] : DC_OOB;
      if (a.mask) {
#pragma unroll
        for (int r = 0; r < 4; ++r) mv[r] = dc_load(rm, off[r]);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) mv[r] = 1.f;
      }
      if (a.res) {
#pragma unroll
        for (int r = 0; r < 4; ++r) rv[r] = dc_load(rr, off[r]);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) rv[r] = 0.f;
      }
      if (a.accumulate) {
#pragma unroll
        for (int r = 0; r < 4; ++r) av[r] = dc_load(ro, off[r]);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) av[r] = 0.f;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = acc[i][j][r] + bv[r];
        v = __builtin_fmaf(v, mv[r] > 0.f ? 1.f : mslope, rv[r]) * a.out_scale;
        if (a.act == RTG_ACT_LRELU) v = rtg_lrelu(v, a.act_slope);
        else if (a.act == RTG_ACT_TANH) v = tanhf(v);
        v += av[r];
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ro, off[r], 0, 0);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------- host
struct DShape {
  int rw16, wb;
};
constexpr DShape kShapes[] = {{2, 4}, {1, 8}, {1, 4}, {2, 8}};       // code digit 1..4: rows per block 128, 128, 64, 256
constexpr int kNT[] = {4, 6, 7, 8};

bool dconv_eligible(const RtgConv1dDesc* d) {
  if (!d->wp16 || d->groups != 1 || d->C2 != 0 || d->out_split != 0 || d->tap_major) return false;
  const int ckc = d->bf16 ? 32 : RTG_CK;             // channels per chunk
  const bool two_d = d->h_k > 1 || d->h_n > 1;
  if (two_d) {
    // forward of the Conv2d layers, backward-data of the row-stride-1 ones; 3 taps along the last axis
    // ... and of the row-strided ones (class-ordered clips, 2 taps of the polyphase walk along the last axis)
    if (d->h_in < 1 || d->h_k < 1 || d->h_n < 1 || d->h_pad < 0 || d->C1 % d->h_k != 0 || d->B % d->h_n != 0) return false;
    if (d->dil != 1 || d->h_stride < 1 || (d->h_mode != 0 && d->h_mode != 1)) return false;
    if (d->h_mode == 1 && (d->C1 / d->h_k) % ckc != 0) return false;      // whole chunks per kernel row
    if (d->h_mode == 1 && d->h_stride > 1) {
      if (d->K != 2 || d->stride != 1 || d->h_stride > 4) return false;
    } else {
      if (d->K != 3 || (d->stride != 1 && d->stride != 2) || d->shuf_S != 1) return false;
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

template <int RW16, int WB, int NT16, int S, int K, bool TWO_D, bool CLS, bool BF, bool HB = CLS>
int launch(const DArgs& a, unsigned blocks, size_t lds_bytes, hipStream_t s) {
  auto k = dconv_kernel<RW16, WB, NT16, S, K, TWO_D, CLS, BF, HB>;
  static std::atomic<unsigned> optin{0};              // (> 64 KB of dynamic LDS: opt-in per kernel and device)
  if (lds_bytes > 64 * 1024 && rtg_lds_optin((const void*)k, optin) != RTG_OK) return RTG_ERANGE;
  RTG_KLAUNCH(k, dim3(blocks), dim3(WB * 64), lds_bytes, s, a);
  return rtg_launch_status();
}

template <int RW16, int WB, int NT16, bool BF>
int launch_sk(const DArgs& a, int S, int K, bool two_d, unsigned blocks, size_t lds_bytes, hipStream_t s) {
  if (two_d) {
    if (a.h_mode == 1 && a.h_stride > 1) return launch<RW16, WB, NT16, 1, 2, true, true, BF>(a, blocks, lds_bytes, s);
    if (S == 1 && K == 3 && a.h_mode == 1) return launch<RW16, WB, NT16, 1, 3, true, false, BF, true>(a, blocks, lds_bytes, s);
    if (S == 1 && K == 3) return launch<RW16, WB, NT16, 1, 3, true, false, BF>(a, blocks, lds_bytes, s);
    if (S == 2 && K == 3 && a.h_mode == 0) return launch<RW16, WB, NT16, 2, 3, true, false, BF>(a, blocks, lds_bytes, s);
    return RTG_EINVAL;
  }
  if (S == 1 && K == 5) return launch<RW16, WB, NT16, 1, 5, false, false, BF>(a, blocks, lds_bytes, s);
  if (S == 3 && K == 5) return launch<RW16, WB, NT16, 3, 5, false, false, BF>(a, blocks, lds_bytes, s);
  if (S == 1 && K == 2) return launch<RW16, WB, NT16, 1, 2, false, false, BF>(a, blocks, lds_bytes, s);
  return RTG_EINVAL;
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
  } c[16];
  int n = 0;
  for (int si = 0; si < 4; ++si) {
    const int mb16 = kShapes[si].rw16 * kShapes[si].wb;
    if (si == 3 && n_mt16 % mb16 != 0) continue;                     // 256-row blocks only where they divide the rows
    const int n_mb = rtg_ceil_div(n_mt16, mb16);
    for (int ni = 0; ni < 4; ++ni) {
      const int BN = kNT[ni] * 16;
      // instances that need more than 256 registers at two waves per SIMD (they spill): 32 rows per wave with >= 6
      // column tiles; 8 column tiles with a strided walk in the 4-wave blocks (twice the staging registers per wave)
      if (kShapes[si].rw16 == 2 && kNT[ni] >= 6) continue;
      const bool two_d = d->h_k > 1 || d->h_n > 1;
      if (kNT[ni] == 8 && (d->stride > 1 || (two_d && kShapes[si].wb == 4))) continue;
      // bf16 (8 staged channels per position, two register sets): 16 rows per wave with 8 column tiles, or with 7 on a
      // strided walk, or with 7 (6 on a strided walk) in the 4-wave blocks
      if (d->bf16 && kShapes[si].rw16 == 1 &&
          (kNT[ni] == 8 || (kNT[ni] == 7 && d->stride > 1) ||
           (kShapes[si].wb == 4 && (kNT[ni] == 7 || (kNT[ni] == 6 && d->stride > 1)))))
        continue;
      const int pw = window_positions((int)(n_cols < BN ? n_cols : BN), d->Q, d->stride, d->K);
      if (2ll * 4 * plane_floats(pw, d->stride) * 4 > 150 * 1024) continue;
      const long long blocks = (long long)n_mb * ((n_cols + BN - 1) / BN);
      // rounds of the chip at one block per CU (two for the 4-wave shapes): the tail round's idle CUs are the loss
      const double slots = 256.0 * (kShapes[si].wb == 4 ? 2 : 1);
      const double rounds = (double)blocks / slots;
      const double eff = rounds / (double)(long long)(rounds + 0.999999);
      const double rows_eff = (double)n_mt16 / (double)(n_mb * mb16);
      const double cols_eff = (double)n_cols / (double)(((n_cols + BN - 1) / BN) * BN);
      c[n].code = RTG_DCONV_CODE + 100 * (si + 1) + kNT[ni];
      c[n].score = eff * rows_eff * cols_eff * (kShapes[si].rw16 == 2 ? 1.0 : 0.95);
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
  if (si < 0 || si > 3) return RTG_EINVAL;
  const int rw16 = kShapes[si].rw16, wb = kShapes[si].wb, BN = nt16 * 16;
  DArgs a;
  // the 16-byte-fragment image follows the standard image of the layer (RtgConv1dDesc.wp16)
  const long long std_size = d->bf16 ? rtg_packed_size_bf16(1, d->Mg, d->Cg, d->K, d->tile_m)
                                     : rtg_packed_size(1, d->Mg, d->Cg, d->K, d->tile_m);
  if (std_size < 0 || (std_size & 3) != 0) return RTG_EINVAL;
  a.x = x; a.wp = wp + std_size; a.bias = bias; a.mask = mask; a.res = res; a.out = out;
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
  a.per_xcd = rtg_ceil_div(total, 8);
  a.PW = window_positions(a.n_cols < BN ? a.n_cols : BN, d->Q, d->stride, d->K);
  a.x_bytes = (d->B / a.h_n) * a.C * a.h_in * d->L_in * 4;       // 1-D: h_n = h_in = 1
  a.out_bytes = d->B * d->out_C * d->out_L * 4;                   // (B = items * h_n)
  const size_t lds_bytes = (size_t)2 * 4 * plane_floats(a.PW, d->stride) * sizeof(float);
  if (lds_bytes > 150 * 1024) return RTG_ERANGE;
  const unsigned blocks = (unsigned)(8 * a.per_xcd);
#define RTG_DC(S_, N_)                                                                                                  \
  if (si == S_ - 1 && nt16 == N_)                                                                                         \
    return d->bf16 ? launch_sk<kShapes[S_ - 1].rw16, kShapes[S_ - 1].wb, N_, true>(a, d->stride, d->K, two_d, blocks, lds_bytes, s) \
                   : launch_sk<kShapes[S_ - 1].rw16, kShapes[S_ - 1].wb, N_, false>(a, d->stride, d->K, two_d, blocks, lds_bytes, s);
  // (32 rows per wave with 6 or more column tiles needs more than 256 registers: never listed, not built)
  RTG_DC(1, 4) RTG_DC(4, 4)
  RTG_DC(2, 4) RTG_DC(2, 6) RTG_DC(2, 7) RTG_DC(2, 8)
  RTG_DC(3, 4) RTG_DC(3, 6) RTG_DC(3, 7) RTG_DC(3, 8)
#undef RTG_DC
  return RTG_EINVAL;
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
