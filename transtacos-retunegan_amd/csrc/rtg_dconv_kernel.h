// rtg_dconv_kernel.h — the dense-layer conv kernel of rtg_dconv.hip (see there) as a template, shared by the translation
// units that instantiate it: rtg_dconv.hip (fp32 and bf16 operands on fp32 tensors) and rtg_dconv_io{1,2,3}.hip (bf16 operands
// on bf16 feature maps in HBM: RtgConv1dDesc.io_bf16).
#pragma once
#include <type_traits>

#include "rtg_common.h"

namespace rtg_dc {


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
  int n_mb, total;            // row blocks, work items
  // XCD x walks the items [xcd_first[x], xcd_first[x + 1]): contiguous ranges of equal WORK — equal counts except for the
  // class-ordered clips, where a tile of residue class c walks only that class's kernel rows (round 5: with equal counts the
  // XCDs that held the classes with more kernel rows ran 4/3 as long as the average and set the launch's duration)
  int xcd_first[9];
  int PW;                     // staged positions per buffer
  int x_bytes, out_bytes;
  // second dimension (RtgConv1dDesc.h_*): a clip is an (item, output row) pair, a channel a (channel, kernel row) pair
  int h_in, h_k, h_stride, h_pad, h_n, h_mode, n_co;
  // class-ordered clips (backward-data over a row-strided layer): output row r only receives kernel rows kh == (r + h_pad)
  // (mod h_stride), so the clip sequence lists the rows of residue class 0 of every item first, then class 1, ...: a
  // column tile inside one class walks only that class's kernel rows.  Class c: first row cls_f, cls_n rows per item,
  // clips [cls_base, ...); cpk = 16-channel chunks per kernel row
  int cls_f[4], cls_n[4], cls_base[4], cpk;
  // bf16 feature maps in HBM (RtgConv1dDesc.io_bf16; instances with IO != 0).  x bf16: the staging walks UNITS of 8
  // consecutive input positions of one clip row (one 16-byte load per channel): xb_ups units per segment (positions
  // 0 .. xb_Lv - 1 of a row are the ones a segment reads), the first segment's walk starts at unit xb_j0 and contributes
  // xb_n0 units, xb_units in all.  mask / res may be bf16 whatever the instance (epilogue loads are per element).
  int xb_ups, xb_Lv, xb_j0, xb_n0, xb_units;
  int mask_b16, res_b16, mask_bytes, res_bytes;
  int xq;                     // bf16 x, 2-D: 0, or 32 = the instance built for rows of >= 32 positions serves this launch
  float enc_slope;            // bf16 output: out = bf16(leaky_relu(v, enc_slope)) — the consumer's activation, applied once
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
// HB: the reduction walks (kernel row, channel) with whole chunks per kernel row — 2-D backward-data (RtgConv1dDesc.h_mode 1)
// and, round 5, the forward on a kernel-row-major weight image (h_mode 2: row = r * h_stride - h_pad + kh, else the same
// code).  One kernel row per chunk makes the row test and the row offset of a staged position a per-chunk quantity and the
// channel a scalar offset of the load; the forward in (channel, kernel row) order (h_mode 0, !HB) pays ~4 vector instructions
// per LOAD for them — measured on the bf16 512 -> 512 3 x 3 layer: 97.7 us, 83.5 with coalesced dummy loads and the address
// arithmetic kept, 58.6 with the arithmetic dropped as well (profiles/r05_dconv_addr_ablations.txt).
// A template parameter although it only selects address arithmetic: with both forms in one loop the compiler's wait-count
// bookkeeping merged their pending loads at every join and waited for the staging loads (and the fragment loads behind
// them) a chunk early
//
// IO (RtgConv1dDesc.io_bf16, round 5: bf16 feature maps in HBM).  Bit 0, x is bf16 NCW and already activated (the producer
// stored bf16(leaky_relu(.)), or x is a gradient): the staging walks UNITS — 8 consecutive positions of one clip row — with
// ONE 16-byte load per (unit, channel) at whatever 2-byte alignment the row has (measured: full rate), a wave task being 8
// units x the 8 channels of one kgrp plane = 64 lanes x 16 bytes.  What a lane holds after the load is 8 positions of one
// channel; the patch wants 8 channels of one position: the wave parks the task in a private [8 channels][64 positions]
// scratch (144-byte rows) and reads it back with two ds_read_b64_tr_b16 — the gfx950 transposing read hands lane i column i
// of a 4-row block — so the transpose costs no vector instruction, and the lane writes ONE position's 16 bytes into the
// plane.  Per 512 staged elements: 1 global load + 2 LDS writes + 2 LDS reads, no conversion, no activation (fp32 tensors:
// 8 loads + ~25 vector instructions + 1 write).  Padding positions are never written: both buffers are zeroed once.
// Bit 1, out is bf16: the epilogue stores bf16(leaky_relu(v, enc_slope)) (enc_slope 1 for gradients).
using u32x4 = unsigned __attribute__((ext_vector_type(4)));
using s16x4 = short __attribute__((ext_vector_type(4)));
// most units a window of `cols` columns touches (rows of >= minq positions): its positions in eighths + two partial units
// per segment it touches; -> wave tasks (plane x block of 8 units) per wave of the block
constexpr int xb_max_units(int cols, int minq, int S, int K) {
  return window_positions(cols, minq, S, K) / 8 + 2 * ((minq >= cols ? 1 : (cols - 2) / minq + 1) + 1) + 1;
}
constexpr int xb_max_tasks(int cols, int minq, int S, int K, int WB) { return (4 * ((xb_max_units(cols, minq, S, K) + 7) / 8) + WB - 1) / WB; }
constexpr int kXbScrF = 8 * 144 / 4;                 // floats of one wave's transposition scratch

// XQ (bf16 x, 2-D, end of round 5): 0, or the shortest row the instance serves.  The task slots of a wave are sized for the
// widest window of the shape, and a window over rows of min_q = 4 positions touches two partial units per row: 9 slots per
// wave for the 96-column 2-tap instance, 150 registers of staging state, spills, two waves per SIMD — where the product's
// rows (the spectrogram discriminators run along the frequency axis: 22-513 positions) need 3.  XQ = 32: 3 slots, 158
// registers, no spills: the class-ordered 64 -> 256 backward-data on bf16 dy 163 -> 108 us (fp32 dy: 131), 32 -> 64 268 -> 156
// (167).  The time of these instances follows the slot count (every slot is a load issue, an LDS round trip and a patch
// write per chunk, against 130-260 matrix cycles): XQ = 16 (4 slots, serves the rows of 22-30 too) measured 115 / 183 us.
template <int RW16, int WB, int NT16, int S, int K, bool TWO_D, bool CLS, bool BF, bool HB, int IO = 0, int XQ = 0>
__global__ __launch_bounds__(WB * 64, 2) void dconv_kernel(const DArgs a) {
  static_assert(!CLS || TWO_D, "class-ordered clips belong to the 2-D backward-data");
  static_assert((!CLS || HB) && (!HB || TWO_D), "h_mode 1 / 2 is 2-D; class-ordered clips are backward-data");
  static_assert(IO == 0 || BF, "bf16 tensors go with bf16 operands");
  constexpr bool XB = (IO & 1) != 0, OB = (IO & 2) != 0;
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
  [[maybe_unused]] const bool fwd2 = HB && !CLS && a.h_mode == 2;      // forward over the kernel-row-major image
  // block -> work item: blocks b and b + 8 share an XCD, each XCD walks a contiguous range of items, the row blocks of
  // one column tile next to each other (they read the same input window: L2 hits)
  const int xcd = (int)(blockIdx.x & 7u);
  const int item = a.xcd_first[xcd] + (int)(blockIdx.x >> 3);
  if (item >= a.xcd_first[xcd + 1]) return;
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
      srow[it] = (!HB || fwd2) ? ho * a.h_stride - a.h_pad : ho + a.h_pad;
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
  constexpr int NSET = BF ? 2 : 1;
  [[maybe_unused]] float st[NSET][NSI][MAXIT];
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
  [[maybe_unused]] unsigned hb_voff[(TWO_D && HB) ? MAXIT : 1];     // (h_mode 1 / 2: per staged position, for the walk's kernel row)
  [[maybe_unused]] int hb_key = -2;                                   // ... which is this one (-1: past the end)
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
  auto stage_issue_f = [&](const Walk& w, auto set_tag) __attribute__((always_inline)) {
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
        // one kernel row per chunk: the rows (and, class-ordered, whether the row's class takes this kernel row) depend on the
        // KERNEL ROW only — computed when the walk enters a kernel row (a uniform branch around vector arithmetic, no memory
        // access inside: the wait counts stay exact), kept for its C / 16 (or 32) chunks
        const int key = is_past ? -1 : w.kh;
        if (key != hb_key) {
          hb_key = key;
          const int dr = is_past ? (1 << 24) : (CLS ? -w.khq : (fwd2 ? w.kh : -w.kh));
#pragma unroll
          for (int it = 0; it < MAXIT; ++it) {
            bool ok = (unsigned)(srow[it] + dr) < (unsigned)a.h_in;
            if constexpr (CLS) ok = ok && scls[it] == w.khr;     // a kernel row of another residue class: zeros
            hb_voff[it] = ok ? soff[it] + (unsigned)dr * chb : DC_OOB;
          }
        }
#pragma unroll
        for (int i = 0; i < NSI; ++i) {
          const int c = w.cw * CKC + (BF ? 8 * skgrp + i : skgrp + 4 * i);
          const unsigned cb = is_past ? 0u : (unsigned)(c * a.h_in) * chb;
#pragma unroll
          for (int it = 0; it < MAXIT; ++it) st[SET][i][it] = dc_load(rx, hb_voff[it], cb);
        }
      }
    } else {
      // (the channel as part of the vector offset — add + and + or per load.  Round 5 tried it as the load's SCALAR offset, no
      // vector arithmetic per load: 2-3 % slower on the 512 -> 512 k5 layers in fp32 and bf16 (157.0 -> 160.1 us, 32.7 -> 33.9);
      // and written as selects on the uniform `is_past` the compiler branched around the loads and waited at every join)
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
  auto stage_write_f = [&](float* buf, auto set_tag) __attribute__((always_inline)) {
    constexpr int SET = decltype(set_tag)::value;
#ifdef RTG_EXP_DC_NOSTWRITE
    return;
#endif
#ifdef RTG_EXP_DC_WAITONLY                              // ablation: the loads are waited for, nothing is converted or written
#pragma unroll
    for (int it = 0; it < MAXIT; ++it)
#pragma unroll
      for (int i = 0; i < NSI; ++i) asm volatile("" ::"v"(st[SET][i][it]));
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

  // ---- x in bf16 (IO bit 0): units of 8 positions, wave tasks of 8 units x 8 channels (see the kernel's header comment)
  constexpr int MAXT = XB ? xb_max_tasks(BN, XQ ? XQ : min_q(K, TWO_D), S, K, WB) : 1;
  [[maybe_unused]] u32x4 xst[NSET][MAXT];
  [[maybe_unused]] unsigned xl_off[MAXT];            // load side: byte offset of this lane's (unit, channel of the plane), or out of range
  [[maybe_unused]] int xl_row[TWO_D ? MAXT : 1];      // 2-D: the unit's clip's input row for kernel row 0
  [[maybe_unused]] int xl_cls[CLS ? MAXT : 1];
  [[maybe_unused]] int xc[(TWO_D && !HB) ? MAXT : 1], xr[(TWO_D && !HB) ? MAXT : 1];   // 2-D forward: this lane's (channel, kernel row)
  [[maybe_unused]] int xs_off[MAXT];                 // store side: float offset of this lane's position in a buffer; < 0: none
  // (a scratch per wave AND task slot: the tasks of a chunk go through write -> transposing read -> write side by side, one
  // LDS round trip for all of them; with one scratch per wave they queued up behind each other's latency)
  [[maybe_unused]] float* const xscr = lds + 2 * bufF + wave * MAXT * kXbScrF;
  [[maybe_unused]] float* const xdump = lds + 2 * bufF + WB * MAXT * kXbScrF + tid * 4;
  [[maybe_unused]] int xb_cnt = 0;                    // task slots of this wave that hold units (wave-uniform)
  [[maybe_unused]] unsigned xb_voff[(XB && TWO_D && HB) ? MAXT : 1];
  [[maybe_unused]] int xb_key = -2;
  [[maybe_unused]] const unsigned chb2 = (unsigned)a.L_in * 2u;
  // scratch addresses: the load side parks 16 bytes at [channel lane & 7][unit slot lane >> 3]; the transposing read of lane
  // 16 g + 4 q + p names row q, columns 16 g + 4 p .. + 3 and hands lane 16 g + i column 16 g + i (rows q = 0..3 as 4 x bf16)
  [[maybe_unused]] float* const xscr_w = xscr + (lane & 7) * 36 + (lane >> 3) * 4;
  [[maybe_unused]] const short* const xscr_r = reinterpret_cast<const short*>(xscr) + ((lane >> 2) & 3) * 72 + (lane >> 4) * 16 + (lane & 3) * 4;
  if constexpr (XB) {
    // both buffers zeroed once: the padding positions (left / right of a row, between two clips' segments) are never staged
    for (int i = tid; i < 2 * planeF; i += WB * 64) reinterpret_cast<f32x4*>(lds)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    // the units of this block's window: segment 0 from unit j0, whole segments, the last one up to the window's end
    const int pos_start = g0 - a.pad;
    const int j0 = pos_start > 0 ? pos_start >> 3 : 0;
    const int G_end = g0 + a.PW;
    const int nseg = (G_end - 1) / a.seg_pw + 1;
    int pe = G_end - (nseg - 1) * a.seg_pw - a.pad;
    pe = pe < a.xb_Lv ? pe : a.xb_Lv;
    const int jhi_last = pe > 0 ? (pe + 7) >> 3 : 0;
    int n0u = (nseg == 1 ? jhi_last : a.xb_ups) - j0;
    n0u = n0u > 0 ? n0u : 0;
    const int total_u = nseg == 1 ? n0u : n0u + (nseg - 2) * a.xb_ups + jhi_last;
    {
      const int n_tasks = 4 * ((total_u + 7) >> 3);   // (plane x block of 8 units), dealt to the waves round robin
      const int c = (n_tasks - wave + WB - 1) / WB;
      xb_cnt = c < 0 ? 0 : (c > MAXT ? MAXT : c);
    }
    [[maybe_unused]] int m0q_b = 0, m0rem_b = 0;
#pragma unroll
    for (int it = 0; it < MAXT; ++it) {
      const int wt = wave + WB * it;
      const int plane = wt & 3;
      const int u = (wt >> 2) * 8 + (lane >> 3), sub = lane & 7;
      int seg = 0, j = j0 + u;
      if (u >= n0u) {
        const int t = u - n0u, sq = t / a.xb_ups;
        seg = 1 + sq;
        j = t - sq * a.xb_ups;
      }
      const int clip = clip0 + seg;
      const bool uok = u < total_u && clip < a.B;
      const int pos0 = 8 * j;
      // store side: position `sub` of the unit
      const int pos = pos0 + sub, o = seg * a.seg_pw + a.pad + pos - g0;
      xs_off[it] = (uok && pos < a.xb_Lv && o >= 0 && o < a.PW) ? plane * planeF + o * 4 : -1;
      // load side: channel 8 * plane + sub of the chunk, positions pos0 .. pos0 + 7
      const int vc0 = 8 * plane + sub;
      if constexpr (TWO_D) {
        int item, ho, cls;
        decode(uok ? clip : 0, item, ho, cls);
        xl_row[it] = (!HB || fwd2) ? ho * a.h_stride - a.h_pad : ho + a.h_pad;
        if constexpr (CLS) {
          xl_row[it] = (ho + a.h_pad - cls) / a.h_stride;
          xl_cls[it] = cls;
        }
        if (!uok) xl_row[it] = -(1 << 28);
        // (item, channel 0, row 0, pos0) — and, backward-data, this lane's channel within the chunk (whole chunks per kernel row)
        xl_off[it] = ((unsigned)item * (unsigned)a.C * (unsigned)a.h_in * (unsigned)a.L_in + (unsigned)pos0) * 2u +
                     (HB ? (unsigned)(vc0 * a.h_in) * chb2 : 0u);
        if constexpr (!HB) {
          xc[it] = vc0 / a.h_k;
          xr[it] = vc0 - xc[it] * a.h_k;
        }
      } else {
        xl_off[it] = uok ? (((unsigned)clip * (unsigned)a.C + (unsigned)vc0) * (unsigned)a.L_in + (unsigned)pos0) * 2u : DC_OOB;
      }
    }
    __syncthreads();
  }
  auto xb_load = [&](unsigned voff, unsigned soff) __attribute__((always_inline)) {
    return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, voff, soff, 0));
  };
  auto stage_issue_b = [&](const Walk& w, auto set_tag) __attribute__((always_inline)) {
    constexpr int SET = decltype(set_tag)::value;
    const bool is_past = w.rc >= a.n_cc;
    if constexpr (TWO_D && !HB) {
      const int q32 = 32 / a.h_k, rem32 = 32 - q32 * a.h_k;
#pragma unroll
      for (int it = 0; it < MAXT; ++it) {
        const int row = xl_row[it] + xr[it];
        const bool ok = !is_past && (unsigned)row < (unsigned)a.h_in;
        xst[SET][it] = xb_load(ok ? xl_off[it] + (unsigned)(xc[it] * a.h_in + row) * chb2 : DC_OOB, 0u);
        const int r2 = xr[it] + rem32;
        const bool wrap = r2 >= a.h_k;
        xr[it] = wrap ? r2 - a.h_k : r2;
        xc[it] += wrap ? q32 + 1 : q32;
      }
    } else if constexpr (TWO_D) {
      const unsigned cb = is_past ? 0u : (unsigned)(w.cw * 32 * a.h_in) * chb2;
      const int key = is_past ? -1 : w.kh;                 // (per kernel row, as in stage_issue_f)
      if (key != xb_key) {
        xb_key = key;
        const int dr = is_past ? (1 << 24) : (CLS ? -w.khq : (fwd2 ? w.kh : -w.kh));
#pragma unroll
        for (int it = 0; it < MAXT; ++it) {
          bool ok = (unsigned)(xl_row[it] + dr) < (unsigned)a.h_in;
          if constexpr (CLS) ok = ok && xl_cls[it] == w.khr;
          xb_voff[it] = ok ? xl_off[it] + (unsigned)(xl_row[it] + dr) * chb2 : DC_OOB;
        }
      }
#pragma unroll
      for (int it = 0; it < MAXT; ++it) xst[SET][it] = xb_load(xb_voff[it], cb);
    } else {
      const unsigned past = is_past ? DC_OOB : 0u;
      const unsigned cb = is_past ? 0u : (unsigned)(w.rc * 32) * chb2;
#pragma unroll
      for (int it = 0; it < MAXT; ++it) xst[SET][it] = xb_load(xl_off[it] | past, cb);
    }
  };
  auto stage_write_b = [&](float* buf, auto set_tag) __attribute__((always_inline)) {
    constexpr int SET = decltype(set_tag)::value;
    // three passes over the wave's task slots — park, read back transposed, write the positions — so that the LDS latencies
    // of the slots overlap; slots past the wave's share hold nothing (uniform skip: LDS traffic only, the loads were issued
    // whatever they fetch so that the vector-memory wait counts stay exact)
#pragma unroll
    for (int it = 0; it < MAXT; ++it) {
      u32x4 v = xst[SET][it];
      asm volatile("" : "+v"(v));                      // the wait for this load sits here, below the multiplications
      if (it < xb_cnt) *reinterpret_cast<u32x4*>(xscr_w + it * kXbScrF) = v;
    }
    asm volatile("" ::: "memory");                     // (same wave, LDS executes in order: no wait between the writes and the reads)
    s16x4 lo[MAXT], hi[MAXT];
#pragma unroll
    for (int it = 0; it < MAXT; ++it) {
      if (it < xb_cnt) {
        lo[it] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(xscr_r + it * (kXbScrF * 2)));
        hi[it] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(xscr_r + it * (kXbScrF * 2) + 4 * 72));
      }
    }
    asm volatile("" ::: "memory");
#pragma unroll
    for (int it = 0; it < MAXT; ++it) {
      if (it < xb_cnt) {
        const unsigned long long l64 = __builtin_bit_cast(unsigned long long, lo[it]), h64 = __builtin_bit_cast(unsigned long long, hi[it]);
        const u32x4 o4 = {(unsigned)l64, (unsigned)(l64 >> 32), (unsigned)h64, (unsigned)(h64 >> 32)};
        float* dst = xs_off[it] >= 0 ? buf + xs_off[it] : xdump;
        *reinterpret_cast<u32x4*>(dst) = o4;
      }
    }
  };
  auto stage_issue = [&](const Walk& w, auto set_tag) __attribute__((always_inline)) {
    if constexpr (XB) stage_issue_b(w, set_tag);
    else stage_issue_f(w, set_tag);
  };
  auto stage_write = [&](float* buf, auto set_tag) __attribute__((always_inline)) {
    if constexpr (XB) stage_write_b(buf, set_tag);
    else stage_write_f(buf, set_tag);
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
  // bf16 (round 5): the weight fragments of a WHOLE chunk are requested CD chunks ahead into a register ring.  At bf16 rates a
  // tap is 8-16 matrix instructions of 16 cycles: fragments requested one tap ahead (the fp32 scheme: a tap is 32
  // instructions of 32 cycles there) arrive from L2 long after the tap before them has finished — the wave sat waiting for
  // its weights.  CD = 1 chunk ahead for 5 taps, 2 for the 2- / 3-tap instances; slot (chunk % CD, tap) is refilled right
  // after the instructions that read it were issued.
  // fp32 (later in round 5): one chunk ahead as well, where the K x RW16 extra fragments cost no occupancy — the 2- / 3-tap
  // instances and the 8-wave blocks (two waves per SIMD whatever they use); measured per family, same box, best block shape:
  // -1 .. -5 % (class-ordered 2-tap 396.7 -> 381.8 us, polyphase 112.8 -> 107.3, 512 -> 512 k5 152.1 -> 147.2), but +10 % on
  // the 5-tap 2-D forward in 4-wave blocks (118 -> 138 registers: three waves per SIMD instead of four), which keeps the
  // one-tap-ahead scheme.  In the train step: config 4 56.95 / 56.76 -> 56.08 / 55.81 ms (two rounds, one box), config 2
  // 26.28 / 26.25 -> 26.43 / 26.31 — the 1-D instances keep the old scheme as well (TWO_D only).  Results are bit-identical
  // either way.
  constexpr int CD = (K >= 5 || RW16 >= 4 || !BF) ? 1 : 2;
#ifdef RTG_EXP_DC_NORING32                               // (A/B: the fp32 instances one tap ahead, as before)
  constexpr bool ARING = BF && CD * RW16 * K <= 12;
#else
  constexpr bool ARING = BF ? CD * RW16 * K <= 12        // (the ring is CD x K x RW16 fragments of 4 registers: 48 at most)
                            : TWO_D && RW16 * K <= 10 && (K <= 3 || WB == 8);
#endif
  [[maybe_unused]] f32x4 ar[ARING ? CD : 1][ARING ? K : 1][RW16];
  auto fetch_a = [&](int slot, int t, int s) __attribute__((always_inline)) {
    const int sc = s < n_steps ? s : n_steps - 1;
#ifndef RTG_EXP_DC_NOA
#pragma unroll
    for (int i = 0; i < RW16; ++i) ar[slot][t][i] = aptr[i][(size_t)sc * 64];
#else
#pragma unroll
    for (int i = 0; i < RW16; ++i) ar[slot][t][i] = f32x4{1.f, 1.f, 1.f, 1.f};
#endif
  };
  auto fetch_b = [&](Frag& f, const float* bsrc) __attribute__((always_inline)) {
#ifndef RTG_EXP_DC_NOB
#pragma unroll
    for (int j = 0; j < NT16; ++j) f.b[j] = *reinterpret_cast<const f32x4*>(bsrc + bcol[j]);
#endif
  };
  auto mma_ring = [&](int slot, int t, const Frag& f) __attribute__((always_inline)) {
#ifdef RTG_EXP_DC_NOMMA
    return;
#endif
    if constexpr (BF) {
#pragma unroll
      for (int i = 0; i < RW16; ++i)
#pragma unroll
        for (int j = 0; j < NT16; ++j)
          asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i][j]) : "v"(ar[slot][t][i]), "v"(f.b[j]));
    } else {
#pragma unroll
      for (int kq = 0; kq < 4; ++kq)
#pragma unroll
        for (int i = 0; i < RW16; ++i)
#pragma unroll
          for (int j = 0; j < NT16; ++j) {
            const float av = ar[slot][t][i][kq], bv = f.b[j][kq];
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i][j]) : "v"(av), "v"(bv));
          }
    }
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
  if constexpr (ARING) {
#pragma unroll
    for (int t = 0; t < K; ++t) fetch_a(0, t, rc0.rc * K + t);
    if constexpr (CD == 2) {
#pragma unroll
      for (int t = 0; t < K; ++t) fetch_a(1, t, rc1.rc * K + t);
    }
    fetch_b(f0, lds);
  } else {
    fetch(f0, rc0.rc * K, lds);
  }

  // one chunk (virtual index v): K taps; `cur` holds the fragments of tap 0 on entry, and of the next chunk's tap 0 on
  // exit (in `cur` again when K is even, in `oth` when K is odd: the caller alternates)
  // `nset`: the register set that holds the next chunk's patch (and takes the request issued at the end of this chunk)
  auto chunk = [&](int v, Frag& cur, Frag& oth, auto nset, auto slot_tag) __attribute__((always_inline)) {
    constexpr int SLOT = decltype(slot_tag)::value;
    const float* bufc = lds + (v & 1) * bufF;
    float* bufn = lds + ((v + 1) & 1) * bufF;
#pragma unroll
    for (int t = 0; t < K; ++t) {
      Frag& fc = (t & 1) ? oth : cur;
      Frag& fn = (t & 1) ? cur : oth;
      // request the next step's fragments, THEN (last tap) the patch of the chunk after the next: the wait for the
      // fragments one step later does not include the patch loads (vmcnt retires in order)
      if constexpr (ARING) {
        if (t + 1 < K) fetch_b(fn, bufc + (t + 1) * 4);
        else fetch_b(fn, bufn);
      } else {
        if (t + 1 < K) fetch(fn, rc0.rc * K + t + 1, bufc + (t + 1) * 4);
        else fetch(fn, rc1.rc * K, bufn);
      }
      if (t == K - 1) stage_issue(NSET == 2 ? rc3 : rc2, nset);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (ARING) {
        mma_ring(SLOT, t, fc);
        __builtin_amdgcn_sched_barrier(0);
        fetch_a(SLOT, t, (CD == 1 ? rc1.rc : rc2.rc) * K + t);       // this slot again CD chunks on
      } else {
        mma(fc);
      }
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
    using Slot0 = std::integral_constant<int, 0>;
    using Slot1 = std::integral_constant<int, (ARING && CD == 2) ? 1 : 0>;
    do {                                    // (n_vp >= 2)
      chunk(cc, f0, f1, Set1{}, Slot0{});
      if constexpr (K & 1) chunk(cc + 1, f1, f0, Set0{}, Slot1{});
      else chunk(cc + 1, f0, f1, Set0{}, Slot1{});
      cc += 2;
    } while (cc < n_vp);
  } else {
    for (; cc < n_v; ++cc) chunk(cc, f0, f1, Set0{}, Set0{});
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
  const rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc((void*)(a.mask ? a.mask : a.out), 0, a.mask ? a.mask_bytes : 0, 0x00020000);
  const rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void*)(a.res ? a.res : a.out), 0, a.res ? a.res_bytes : 0, 0x00020000);
  // element sizes: out by the instance, mask / res by the descriptor (a bf16 feature map as leaky-relu mask, a bf16 gradient as
  // residual); offsets are computed in elements
  constexpr unsigned OES = OB ? 2u : 4u;
  auto ld_elem = [&](rsrc_t r, unsigned eoff, bool ok, bool b16) __attribute__((always_inline)) {
    if (b16) {
      const unsigned short h = (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(r, ok ? eoff * 2u : DC_OOB, 0, 0);
      return __builtin_bit_cast(float, (unsigned)h << 16);
    }
    return dc_load(r, ok ? eoff * 4u : DC_OOB);
  };
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
      rowoff[r] = (unsigned)(ch * a.h_n * a.out_L + ph);               // (elements; h_n == 1 in 1-D)
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
        col = ((unsigned)(item * a.out_C * a.h_n + ho) * (unsigned)a.out_L + (unsigned)(q * So));
      } else {
        col = ((unsigned)(clip * a.out_C) * (unsigned)a.out_L + (unsigned)(q * So));
      }
      unsigned off[4];                                  // element offsets
      bool ok[4];
      float mv[4], rv[4], av[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        ok[r] = (unsigned)(qs + rowph[r]) < (unsigned)a.out_L;
        off[r] = col + rowoff[r];
      }
      if (a.mask) {
#pragma unroll
        for (int r = 0; r < 4; ++r) mv[r] = ld_elem(rm, off[r], ok[r], a.mask_b16 != 0);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) mv[r] = 1.f;
      }
      if (a.res) {
#pragma unroll
        for (int r = 0; r < 4; ++r) rv[r] = ld_elem(rr, off[r], ok[r], a.res_b16 != 0);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) rv[r] = 0.f;
      }
      if (!OB && a.accumulate) {
#pragma unroll
        for (int r = 0; r < 4; ++r) av[r] = dc_load(ro, ok[r] ? off[r] * 4u : DC_OOB);
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
        if constexpr (OB) {
          // the consumer's activation once, here, then round to nearest even: what the next layer stages is what it multiplies
          const __bf16 h = (__bf16)rtg_lrelu(v, a.enc_slope);
          __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(short, h), ro, ok[r] ? off[r] * OES : DC_OOB, 0, 0);
        } else {
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ro, ok[r] ? off[r] * 4u : DC_OOB, 0, 0);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------- launch
struct DShape {
  int rw16, wb;
};
// code digit 1..5: rows per block 128, 128, 64, 256, 256.  Shape 5 (bf16 only, round 5): 64 rows per wave — at bf16 matrix
// rates the kernel lives on operand delivery (a 16x16x32 instruction eats 1 KB of weights from L1 and 1 KB of patch from LDS
// every 16 cycles: at 32 rows x 64 columns per wave that is 64 B/clk/CU of L1 and 128 B/clk/CU of LDS at full rate — both
// their limits); 64 rows per wave halve the LDS side
constexpr DShape kShapes[] = {{2, 4}, {1, 8}, {1, 4}, {2, 8}, {4, 4}};
constexpr int kNumShapes = 5;
constexpr int kNT[] = {4, 6, 7, 8};

// Which (block shape, column tiles, geometry, arithmetic, tensor types) instances EXIST.  One rule for the candidate list
// (rtg_dconv_candidates) and for the dispatch below: a shape the list does not hold is not compiled, and rtg_conv1d refuses a
// caller-fixed code that is not on the list (round 6; rounds 3-5 built every combination — 672 instances, 70 of them
// with scratch, for the 68 a step launches).  Left out:
//   * 32 / 64 rows per wave with 6 or more column tiles, and 8 column tiles with a strided walk or in the 4-wave 2-D blocks:
//     more than 256 registers at two waves per SIMD (they spill);
//   * 64 rows per wave in fp32 (the shape exists for the bf16 operand-delivery limit, section "launch" below);
//   * bf16 operands, 16 rows per wave (8 staged channels per position, two register sets): 8 column tiles, 7 on a strided
//     walk, 7 (6 on a strided walk) in the 4-wave blocks;
//   * bf16 TENSORS (IO != 0): 64-column tiles and 16 / 32 rows per wave only (what the tuner picked in every measured step,
//     profiles/r05a_*; the strided 2-D walks at 64 rows per wave spill 1-2 registers), and no 2-D walk over the channel-major image (h_mode 0: the forward runs over the kernel-row-major one, rtg/bank.py FWD_KH_MAJOR;
//     a shape without native instance goes through fp32 copies, rtg/ops.py).
constexpr bool dc_built(int rw16, int wb, int nt16, int S, bool two_d, bool hb, bool bf, int io) {
  if (rw16 >= 2 && nt16 >= 6) return false;
  if (rw16 == 4 && !bf) return false;
  if (nt16 == 8 && (S > 1 || (two_d && wb == 4))) return false;
  if (bf && rw16 == 1 && (nt16 == 8 || (nt16 == 7 && S > 1) || (wb == 4 && (nt16 == 7 || (nt16 == 6 && S > 1))))) return false;
  if (io != 0 && (nt16 != 4 || rw16 == 4 || (two_d && !hb))) return false;
  return true;
}

// dynamic LDS of an instance: two patch buffers; bf16 input: + a transposition scratch per wave and a dump slot per lane
inline size_t lds_bytes_for(int PW, int S, int WB, int xb_tasks) {
  return (size_t)2 * 4 * plane_floats(PW, S) * sizeof(float) +
         (xb_tasks ? (size_t)WB * xb_tasks * kXbScrF * 4 + (size_t)WB * 64 * 16 : 0);
}

template <int RW16, int WB, int NT16, int S, int K, bool TWO_D, bool CLS, bool BF, bool HB = CLS, int IO = 0, int XQ = 0>
int launch(const DArgs& a, unsigned blocks, size_t lds_bytes, hipStream_t s) {
  auto k = dconv_kernel<RW16, WB, NT16, S, K, TWO_D, CLS, BF, HB, IO, XQ>;
  if constexpr ((IO & 1) != 0) {
    // every unit of the widest window of this problem must find a task slot (a unit left out would be a patch of zeros)
    const int nseg = (a.PW + a.seg_pw - 2) / a.seg_pw + 1;
    if (a.PW / 8 + 2 * nseg + 1 > 2 * WB * xb_max_tasks(NT16 * 16, XQ ? XQ : min_q(K, TWO_D), S, K, WB)) return RTG_ERANGE;
    if (XQ != 0 && a.Q < XQ) return RTG_ERANGE;
  }
  static std::atomic<unsigned> optin{0};              // (> 64 KB of dynamic LDS: opt-in per kernel and device)
  if (lds_bytes > 64 * 1024 && rtg_lds_optin((const void*)k, optin) != RTG_OK) return RTG_ERANGE;
  RTG_KLAUNCH(k, dim3(blocks), dim3(WB * 64), lds_bytes, s, a);
  return rtg_launch_status();
}

template <int RW16, int WB, int NT16, int S, int K, bool TWO_D, bool CLS, bool BF, bool HB = CLS, int IO = 0, int XQ = 0>
int launch_if_built(const DArgs& a, unsigned blocks, size_t lds_bytes, hipStream_t s) {
  if constexpr (dc_built(RW16, WB, NT16, S, TWO_D, HB, BF, IO)) return launch<RW16, WB, NT16, S, K, TWO_D, CLS, BF, HB, IO, XQ>(a, blocks, lds_bytes, s);
  else return RTG_EINVAL;
}

// the 2-D instances that walk (kernel row, channel): backward-data, and the forward over the kernel-row-major image
template <int RW16, int WB, int NT16, bool BF, int IO, int XQ>
int launch_hb(const DArgs& a, int S, int K, unsigned blocks, size_t lds_bytes, hipStream_t s) {
  if (a.h_mode == 2) {
    if (S == 1 && K == 3) return launch_if_built<RW16, WB, NT16, 1, 3, true, false, BF, true, IO, XQ>(a, blocks, lds_bytes, s);
    if (S == 2 && K == 3) return launch_if_built<RW16, WB, NT16, 2, 3, true, false, BF, true, IO, XQ>(a, blocks, lds_bytes, s);
    if (S == 3 && K == 5) return launch_if_built<RW16, WB, NT16, 3, 5, true, false, BF, true, IO, XQ>(a, blocks, lds_bytes, s);
    return RTG_EINVAL;
  }
  if (a.h_mode == 1 && a.h_stride > 1) return launch_if_built<RW16, WB, NT16, 1, 2, true, true, BF, true, IO, XQ>(a, blocks, lds_bytes, s);
  if (S == 1 && K == 3 && a.h_mode == 1) return launch_if_built<RW16, WB, NT16, 1, 3, true, false, BF, true, IO, XQ>(a, blocks, lds_bytes, s);
  return RTG_EINVAL;
}

template <int RW16, int WB, int NT16, bool BF, int IO = 0>
int launch_sk(const DArgs& a, int S, int K, bool two_d, unsigned blocks, size_t lds_bytes, hipStream_t s) {
  if (two_d) {
    if (a.h_mode == 2 || a.h_mode == 1) {
      if constexpr ((IO & 1) != 0) {
        if (a.xq == 32) return launch_hb<RW16, WB, NT16, BF, IO, 32>(a, S, K, blocks, lds_bytes, s);
      }
      return launch_hb<RW16, WB, NT16, BF, IO, 0>(a, S, K, blocks, lds_bytes, s);
    }
    if (S == 1 && K == 3) return launch_if_built<RW16, WB, NT16, 1, 3, true, false, BF, false, IO>(a, blocks, lds_bytes, s);
    if (S == 2 && K == 3 && a.h_mode == 0) return launch_if_built<RW16, WB, NT16, 2, 3, true, false, BF, false, IO>(a, blocks, lds_bytes, s);
    // (round 5: StftDiscriminator along the frequency axis — its (5, 3) kernels with stride (3, 2) walk 5 taps at stride 3)
    if (S == 3 && K == 5 && a.h_mode == 0) return launch_if_built<RW16, WB, NT16, 3, 5, true, false, BF, false, IO>(a, blocks, lds_bytes, s);
    return RTG_EINVAL;
  }
  if (S == 1 && K == 5) return launch_if_built<RW16, WB, NT16, 1, 5, false, false, BF, false, IO>(a, blocks, lds_bytes, s);
  if (S == 3 && K == 5) return launch_if_built<RW16, WB, NT16, 3, 5, false, false, BF, false, IO>(a, blocks, lds_bytes, s);
  if (S == 1 && K == 2) return launch_if_built<RW16, WB, NT16, 1, 2, false, false, BF, false, IO>(a, blocks, lds_bytes, s);
  return RTG_EINVAL;
}

// (shape index 0..3, 16-column tiles) -> instance, for one arithmetic / tensor-type combination
template <bool BF, int IO>
int launch_shape(const DArgs& a, int si, int nt16, int S, int K, bool two_d, unsigned blocks, size_t lds_bytes, hipStream_t s) {
#define RTG_DC(S_, N_) \
  if (si == S_ - 1 && nt16 == N_) return launch_sk<kShapes[S_ - 1].rw16, kShapes[S_ - 1].wb, N_, BF, IO>(a, S, K, two_d, blocks, lds_bytes, s);
  // (32 rows per wave with 6 or more column tiles needs more than 256 registers: never listed, not built)
  RTG_DC(1, 4) RTG_DC(4, 4)
  RTG_DC(2, 4) RTG_DC(2, 6) RTG_DC(2, 7) RTG_DC(2, 8)
  RTG_DC(3, 4) RTG_DC(3, 6) RTG_DC(3, 7) RTG_DC(3, 8)
  if constexpr (BF) {
    RTG_DC(5, 4)
  }
#undef RTG_DC
  return RTG_EINVAL;
}

}  // namespace rtg_dc

// the bf16-operand instances on fp32 tensors (rtg_dconv_bf.hip) and the bf16-tensor instances (rtg_dconv_io{1,2,3}.hip: IO = the
// file's number) live in translation units of their own
int rtg_dconv_launch_bf(const rtg_dc::DArgs& a, int si, int nt16, int S, int K, bool two_d, unsigned blocks, size_t lds_bytes, hipStream_t s);
int rtg_dconv_launch_io1(const rtg_dc::DArgs& a, int si, int nt16, int S, int K, bool two_d, unsigned blocks, size_t lds_bytes, hipStream_t s);
int rtg_dconv_launch_io2(const rtg_dc::DArgs& a, int si, int nt16, int S, int K, bool two_d, unsigned blocks, size_t lds_bytes, hipStream_t s);
int rtg_dconv_launch_io3(const rtg_dc::DArgs& a, int si, int nt16, int S, int K, bool two_d, unsigned blocks, size_t lds_bytes, hipStream_t s);
