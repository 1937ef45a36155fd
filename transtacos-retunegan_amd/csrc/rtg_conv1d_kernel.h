// rtg_conv1d_kernel.h — the implicit-GEMM Conv1d MFMA kernel (see rtg_conv1d.hip for the operator and the mapping).
// Included by one translation unit per (TM, MT, NT) block shape (rtg_conv1d_t*.hip) so that the instances compile in
// parallel; rtg_conv1d.hip holds the host side.
#pragma once
#include <cstdio>
#include <cstdlib>

#include "rtg_common.h"

namespace rtg_cv {

#ifdef RTG_STAMPS
#define RTG_STAMP(i) do { if (a.dbg) a.dbg[((size_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#define RTG_STAMP2(i) do { if (a.dbg && blockIdx.x < 512 && (i) < 128) a.dbg[(4u << 20) + ((size_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) * 128 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define RTG_STAMP(i) do {} while (0)
#define RTG_STAMP2(i) do {} while (0)
#endif

struct ConvArgs {
  const float *x1, *x2, *aux, *wp, *bias, *mask, *res;
  float *out, *out2;
  int out_split;
  int B, C1, C2, L_in, groups, Cg, Mg, K, stride, dil, pad, Q, out_C, out_L, shuf_S, shuf_P;
  int pre_mode;
  float pre_slope, mask_slope, out_scale;
  int act;
  float act_slope;
  int accumulate;
  int n_cc, n_mt, WM, WN, PW, PH, ROW, m_blocks;
  int seg_len, seg_pitch, seg_nb, seg_pw;   // segment packing (seg_len == 0: one clip per block column)
  int seg_dense;                            // packed: the block's columns are a window of the dense (clip, q) sequence
  // 2-D mode (h_k > 1 or h_n > 1): a "clip" is one (batch item, output row) pair and a "channel" one (channel, kernel
  // row) pair; the kernel row picks which input row of the [items, C, h_in, L_in] tensor the patch row comes from
  int two_d, h_in, h_k, h_stride, h_pad, h_n, h_mode;
  int x_bytes, aux_bytes;
  // tap-major K order (few input channels per group): the k-steps walk (channel, group of KK taps) instead of
  // (tap, group of KK channels); `K` then counts groups of CPN k-steps, K_real the taps, tab_off the LDS offset table
  int tapmajor, K_real, TG, tab_off;
  // block -> work mapping (XCD-aware, see the kernel): q tiles, total work items, work items per XCD
  int gx, total, per_xcd;
#ifdef RTG_STAMPS
  unsigned long long* dbg;   // diagnostic builds only (tools/dev_build.sh): per-block s_memtime stamps
#endif
};

// 2-D backward-data over a row-strided convolution (h_mode == 1, h_stride > 1): input row h of dx only receives kernel
// rows kh == (h + h_pad) (mod h_stride).  Blocks are therefore made class-pure: the z index of a block of packed clips
// enumerates (batch item, residue class r, group of seg_nb rows of that class), so that the block walks only the
// (kernel row, channel) chunks of its class instead of multiplying the other chunks with zeros.
struct RowClass {
  int item, cls, first, j0, cnt;     // rows of the block: first + (j0 + seg) * h_stride for seg < cnt
};
__device__ __forceinline__ int rtg_class_first(int r, int s, int pad) {
  int f = (r - pad) % s;
  return f < 0 ? f + s : f;
}
__device__ __forceinline__ RowClass rtg_block_rows(int bz, int H, int s, int pad, int seg_nb) {
  int per_item = 0;
  for (int r = 0; r < s; ++r) {
    const int f = rtg_class_first(r, s, pad);
    per_item += f < H ? ((H - f + s - 1) / s + seg_nb - 1) / seg_nb : 0;
  }
  RowClass rc;
  rc.item = bz / per_item;
  int rem = bz - rc.item * per_item;
  rc.cls = 0; rc.first = 0; rc.j0 = 0; rc.cnt = 0;
  for (int r = 0; r < s; ++r) {
    const int f = rtg_class_first(r, s, pad);
    const int n = f < H ? (H - f + s - 1) / s : 0;
    const int nb = (n + seg_nb - 1) / seg_nb;
    if (rem < nb) {
      rc.cls = r; rc.first = f; rc.j0 = rem * seg_nb;
      rc.cnt = n - rc.j0 < seg_nb ? n - rc.j0 : seg_nb;
      break;
    }
    rem -= nb;
  }
  return rc;
}

template <int TM>
struct Mfma;
template <>
struct Mfma<32> {
  using acc_t = f32x16;
  static constexpr int NREG = 16;
  static __device__ __forceinline__ acc_t run(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int row(int lane, int r) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }
};
template <>
struct Mfma<16> {
  using acc_t = f32x4;
  static constexpr int NREG = 4;
  static __device__ __forceinline__ acc_t run(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int row(int lane, int r) { return (lane >> 4) * 4 + r; }
};

// Global -> register staging of RPW patch rows of channel chunk `cc` through buffer loads: out-of-range elements
// (zero padding, channels past Cg, other clips' rows) get an offset beyond num_records, for which the hardware returns
// 0 without touching memory — no branches, no clamping, all loads of a chunk issue back to back.
using rsrc_t = __amdgpu_buffer_rsrc_t;
#define RTG_OOB 0x80000000u

__device__ __forceinline__ float buf_load(rsrc_t r, unsigned off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
}

template <bool AUX, int RPW, int MAXIT>
__device__ __forceinline__ void stage_rows(const ConvArgs& a, rsrc_t r1, rsrc_t r2, rsrc_t raux,
                                           float (&st)[RPW][MAXIT], const unsigned (&eb)[MAXIT],
                                           const unsigned (&epos)[MAXIT], const int (&ehq)[MAXIT],
                                           const int (&ehr)[MAXIT], int cc, int wave, int g, float slope,
                                           bool aux_tanh) {
  if (a.two_d) {
    // 2-D: virtual channel c = (ci, kh); the patch row of clip (item, row) comes from input row hq +- kh/stride
    const int cin = a.C1 / a.h_k;
    const unsigned item_bytes = (unsigned)cin * (unsigned)a.h_in * (unsigned)a.L_in * 4u;
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const int c = cc * RTG_CK + wave * RPW + i;
      int ci, kh;
      if (a.h_mode == 0) { ci = c / a.h_k; kh = c - ci * a.h_k; }     // forward: virtual channel = (channel, kernel row)
      else { kh = c / cin; ci = c - kh * cin; }                       // backward-data: (kernel row, channel)
      int khq, khr, sgn;
      if (a.h_mode == 0) { khq = kh; khr = 0; sgn = 1; }                       // forward: row = ho*s - p + kh
      else { khq = kh / a.h_stride; khr = kh - khq * a.h_stride; sgn = -1; }   // backward-data: (h + p - kh) / s
      const unsigned rowoob = (c < a.Cg) ? 0u : RTG_OOB;
#pragma unroll
      for (int it = 0; it < MAXIT; ++it) {
        const int hrow = ehq[it] + sgn * khq;
        const bool hok = ehr[it] == khr && hrow >= 0 && hrow < a.h_in;
        const unsigned offs = hok ? (eb[it] * item_bytes + (unsigned)(ci * a.h_in + hrow) * (unsigned)a.L_in * 4u +
                                     (epos[it] & ~RTG_OOB)) | (epos[it] & RTG_OOB) | rowoob
                                  : RTG_OOB;
        float v = buf_load(r1, offs);
        if (AUX) {
          const float av = buf_load(raux, offs);
          v *= aux_tanh ? (1.f - av * av) : (av > 0.f ? 1.f : slope);
        }
        st[i][it] = v;
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < RPW; ++i) {
    const int c = cc * RTG_CK + wave * RPW + i;
    const int gc = g * a.Cg + c;
    const bool in1 = gc < a.C1;
    const rsrc_t r = in1 ? r1 : r2;
    const unsigned cstride = (unsigned)(in1 ? a.C1 : a.C2) * (unsigned)a.L_in * 4u;   // bytes per clip
    const unsigned rowoff = (unsigned)(in1 ? gc : gc - a.C1) * (unsigned)a.L_in * 4u;
    const unsigned rowoob = (c < a.Cg) ? 0u : RTG_OOB;
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
      // valid byte offsets stay below 2^31 (host-checked); the OOB bit is OR-ed in, never added
      const unsigned offs = (eb[it] * cstride + rowoff + (epos[it] & ~RTG_OOB)) | (epos[it] & RTG_OOB) | rowoob;
      float v = buf_load(r, offs);
      if (AUX) {
        const float av = buf_load(raux, offs);
        v *= aux_tanh ? (1.f - av * av) : (av > 0.f ? 1.f : slope);
      }
      st[i][it] = v;       // the leaky-relu of the plain path is applied when the tile is written to LDS, so that
                           // nothing here waits for the loads: they stay in flight under the MFMA loop
    }
  }
}

// Up to RTG_MAX_GROUP independent convolutions in ONE launch (rtg_conv1d_group): the sub-discriminators of a stack run
// the same layer shape family on different clips with different weights, and each alone often gives a CU less than one
// block; side by side they fill the chip and share one launch.  blk_end[i] = first block after problem i.
struct GroupArgs {
  int n;
  unsigned blk_end[RTG_MAX_GROUP];
  ConvArgs p[RTG_MAX_GROUP];
};

// MAXIT: 64-float pieces of a patch row each lane stages (>= PW / 64).  CLS: strided 2-D backward-data with class-pure
// blocks (RowClass) — a compile-time flag so that every other launch carries none of that bookkeeping.
// BF: bf16 operands (activations rounded when they are written to LDS, weights packed as bf16 fragments), multiplied on
// the bf16 matrix cores with fp32 accumulation — everything around the operands (geometry, staging loads, epilogue) is
// shared with the fp32 path.
typedef short bf4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

template <int TM>
struct MfmaBf;
template <>
struct MfmaBf<32> {
  static __device__ __forceinline__ f32x16 run(bf4 a, bf4 b, f32x16 c) {
#ifdef RTG_EXP_NOMFMA_BF
    c[0] += (float)(a.x + b.x);      // ablation: operands stay live, no matrix instruction
    return c;
#endif
    return __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a, b, c, 0, 0, 0);
  }
};
template <>
struct MfmaBf<16> {
  static __device__ __forceinline__ f32x4 run(bf4 a, bf4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
  }
};

template <int TM, int MT, int NT, int MAXIT, bool CLS, bool BF>
__device__ __forceinline__ void conv1d_mfma_body(const ConvArgs& a, const unsigned bid) {
  using M = Mfma<TM>;
  using acc_t = typename M::acc_t;
  constexpr int KK = 64 / TM;           // K-values consumed per MFMA
  constexpr int CPN = RTG_CK / KK;      // MFMA k-steps per (chunk, tap)
  constexpr int RPW = RTG_CK / 4;       // patch rows staged per wave

  extern __shared__ __attribute__((aligned(16))) float lds[];

  const int tid = threadIdx.x, lane = tid & 63;
  RTG_STAMP(0);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / a.WN, wn = wave - wm * a.WN;
  // Work item of this block.  Workgroups are dealt round-robin over the 8 XCDs (block b and b+8 share an XCD and its
  // L2), so block b takes item (b % 8) * per_xcd + b / 8: each XCD walks a CONTIGUOUS range of items, and the items are
  // ordered with the row blocks of one input patch next to each other (m block fastest, then group, q tile, clip) —
  // the m blocks that re-read the same patch run on the same XCD at about the same time and hit its L2 instead of
  // fetching the patch once per row block from HBM.  (Speed only: any placement computes the same result.)
#ifdef RTG_EXP_EMPTY
  return;
#endif
  const int item = (int)(bid & 7u) * a.per_xcd + (int)(bid >> 3);
  if (item >= a.total) return;
  int wrk = item;
  const int mb = wrk % a.m_blocks;
  wrk /= a.m_blocks;
  const int g = wrk % a.groups;
  wrk /= a.groups;
  const int bx = wrk % a.gx, bz = wrk / a.gx;
  const int mt0 = (mb * a.WM + wm) * MT;
  const int BN = a.WN * NT * TM;
  const bool packed = a.seg_len > 0;
  int b0 = packed ? bz * a.seg_nb : bz;         // first clip of this block
  int q0 = 0;                                   // dense window: position of the block's first column within clip b0
  if (packed && a.seg_dense) {
    const int c0 = bz * BN;
    b0 = c0 / a.Q;
    q0 = c0 - b0 * a.Q;
  }
  constexpr bool cls_mode = CLS;        // host: two_d && h_mode == 1 && h_stride > 1
  RowClass rc = {0, 0, 0, 0, 0};
  if (cls_mode) {
    if (packed) {
      rc = rtg_block_rows(bz, a.h_n, a.h_stride, a.h_pad, a.seg_nb);
    } else {                                     // one clip (row) per block
      rc.item = b0 / a.h_n;
      rc.first = b0 - rc.item * a.h_n;
      rc.cls = (rc.first + a.h_pad) % a.h_stride;
      rc.cnt = 1;
    }
  }
  const int q_blk = packed ? 0 : bx * BN;
  const int o_start = q_blk * a.stride - a.pad;
  const int bufsz = RTG_CK * a.ROW;

  // ---- staging geometry (per thread, independent of the channel chunk): LDS offset, clip index and byte position
  // within the row (with the out-of-bounds bit set for zero padding / other clips)
  int loff[MAXIT], ehq[MAXIT], ehr[MAXIT];
  unsigned epos[MAXIT], eb[MAXIT];
#pragma unroll
  for (int it = 0; it < MAXIT; ++it) {
    const int o = lane + 64 * it;
    loff[it] = o;
    if (a.stride != 1) loff[it] = (o % a.stride) * a.PH + o / a.stride;   // uniform branch: no division for stride 1
    int pos, bb;
    bool ok = o < a.PW;
    int seg = 0;
    if (packed) {
      seg = o / a.seg_pitch;
      const int w = o - seg * a.seg_pitch;
      bb = b0 + seg;
      pos = w - a.pad;
      ok = ok && seg < a.seg_nb && w < a.seg_pw && (cls_mode ? seg < rc.cnt : bb < a.B);
    } else {
      bb = b0;
      pos = o_start + o;
    }
    ok = ok && pos >= 0 && pos < a.L_in;
    epos[it] = ok ? (unsigned)pos * 4u : RTG_OOB;
    eb[it] = ok ? (unsigned)bb : 0u;
    ehq[it] = 0;
    ehr[it] = 0;
    if (a.two_d) {                       // clip -> (batch item, row); see stage_rows
      int item = (int)eb[it] / a.h_n;
      int hh = (int)eb[it] - item * a.h_n;
      if (cls_mode) {
        item = ok ? rc.item : 0;
        hh = rc.first + (rc.j0 + seg) * a.h_stride;
      }
      eb[it] = (unsigned)item;
      if (a.h_mode == 0) {
        ehq[it] = hh * a.h_stride - a.h_pad;
      } else {
        ehq[it] = (hh + a.h_pad) / a.h_stride;
        ehr[it] = (hh + a.h_pad) - ehq[it] * a.h_stride;
      }
    }
  }
  float st[RPW][MAXIT];
  const bool aux_tanh = a.pre_mode == RTG_PRE_MUL_DTANH;
  const float slope = (a.pre_mode == RTG_PRE_NONE) ? 1.f : a.pre_slope;
  const float wslope = (a.pre_mode == RTG_PRE_LRELU) ? a.pre_slope : 1.f;   // applied at the LDS write
  const rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc((void*)a.x1, 0, a.x_bytes, 0x00020000);
  const rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x2 ? a.x2 : a.x1), 0,
                                                      a.x2 ? a.B * a.C2 * a.L_in * 4 : 0, 0x00020000);
  const rsrc_t raux = __builtin_amdgcn_make_buffer_rsrc((void*)(a.aux ? a.aux : a.x1), 0, a.aux_bytes, 0x00020000);

  auto stage = [&](int cc) __attribute__((always_inline)) {
    if (a.aux) stage_rows<true, RPW, MAXIT>(a, r1, r2, raux, st, eb, epos, ehq, ehr, cc, wave, g, slope, aux_tanh);
    else stage_rows<false, RPW, MAXIT>(a, r1, r2, raux, st, eb, epos, ehq, ehr, cc, wave, g, slope, aux_tanh);
  };
  auto swrite = [&](float* buf) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      float* rowp = buf + (wave * RPW + i) * a.ROW;
#pragma unroll
      for (int it = 0; it < MAXIT; ++it)
        if (lane + 64 * it < a.PW) {
          float v = st[i][it];
          // keep the consumption of the prefetched values BELOW the MFMA loop: without this the compiler hoists the
          // activation (and with it the s_waitcnt for the loads) above the loop and the prefetch hides nothing
          asm volatile("" : "+v"(v) : : "memory");
          rowp[loff[it]] = v > 0.f ? v : v * wslope;
        }
    }
  };

  // bf16: the wave's four channel rows of one position become ONE 8-byte LDS element [channel group = wave][position]
  // (the 4 consecutive K values a lane of the bf16 MFMA holds), written with a single ds_write_b64
  auto swrite_bf = [&](float* buf) __attribute__((always_inline)) {
    static_assert(RPW == 4, "one 4-channel group per wave");
    u32x2* rowp = reinterpret_cast<u32x2*>(buf) + wave * a.ROW;
#pragma unroll
    for (int it = 0; it < MAXIT; ++it)
      if (lane + 64 * it < a.PW) {
        unsigned short hb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float v = st[i][it];
          asm volatile("" : "+v"(v) : : "memory");
          v = v > 0.f ? v : v * wslope;
          hb[i] = __builtin_bit_cast(unsigned short, (__bf16)v);
        }
        u32x2 pk;
        pk.x = (unsigned)hb[0] | ((unsigned)hb[1] << 16);
        pk.y = (unsigned)hb[2] | ((unsigned)hb[3] << 16);
        rowp[loff[it]] = pk;
      }
  };

  // ---- accumulators
  acc_t acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < M::NREG; ++r) acc[i][j][r] = 0.f;

  // ---- operand addressing
  const int n_lane = lane & (TM - 1), kk = lane / TM;
  const int bbase = kk * a.ROW + wn * NT * TM + n_lane;
  // packed clips: the block's columns enumerate (clip, q) densely — column n is output q = n % Q of clip n / Q — while a
  // clip's patch occupies seg_len virtual positions in LDS (its outputs plus the halo gap): pcol[j] is the LDS position
  // of this lane's column in tile j (junk columns past the last clip read position 0 and are dropped in the epilogue)
  int pcol[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = q0 + (wn * NT + j) * TM + n_lane;
    const int seg = packed ? n / a.Q : 0;
    pcol[j] = kk * a.ROW + (seg < a.seg_nb ? seg * a.seg_len + (n - seg * a.Q) : 0);
  }
  const float* wptr[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    int mt = mt0 + i;
    if (mt > a.n_mt - 1) mt = a.n_mt - 1;   // clamped duplicate tile, discarded in the epilogue
    wptr[i] = a.wp + ((size_t)(g * a.n_mt + mt) * a.n_cc) * a.K * (RTG_CK * TM) + lane;
  }
  // chunks this block walks: all of them, or (strided 2-D backward-data, rows of one residue class, see RowClass)
  // only the kernel rows of that class — virtual chunk v -> real chunk real_cc(v)
  int n_cc = a.n_cc, sub_cpk = 0, sub_kh0 = 0;
  if constexpr (cls_mode) if (((a.C1 / a.h_k) % RTG_CK) == 0) {
    sub_cpk = (a.C1 / a.h_k) / RTG_CK;
    sub_kh0 = rc.cls;
    const int nkh = rc.cls < a.h_k ? (a.h_k - rc.cls + a.h_stride - 1) / a.h_stride : 0;
    n_cc = nkh * sub_cpk;
  }
  auto real_cc = [&](int v) __attribute__((always_inline)) {
    if constexpr (!cls_mode) return v;
    else return sub_cpk ? (sub_kh0 + (v / sub_cpk) * a.h_stride) * sub_cpk + v % sub_cpk : v;
  };
#ifdef RTG_EXP_SKIPLOOP
  const int n_steps = 0;
#else
  const int n_steps = n_cc * a.K;
#endif
  const size_t wstep = (size_t)RTG_CK * TM;      // floats per (chunk, tap) step of one m tile
  size_t wofs = (size_t)real_cc(0) * a.K * wstep;   // offset of the step whose A fragments were requested last
  if constexpr (BF) {
    // ---------------- bf16 main loop: NMF MFMAs per (chunk, tap) step (8 or 16 channels each), same pipeline as below
    constexpr int NMF = TM == 32 ? 2 : 1;
    const u32x2* wpb[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      int mt = mt0 + i;
      if (mt > a.n_mt - 1) mt = a.n_mt - 1;
      wpb[i] = reinterpret_cast<const u32x2*>(a.wp) + ((size_t)(g * a.n_mt + mt) * a.n_cc) * a.K * (NMF * 64) + lane;
    }
    bf4 A0[MT][NMF], A1[MT][NMF];
    // offset (in 8-byte fragments) of the step whose A fragments were requested last; class-pure blocks (strided 2-D
    // backward-data) walk only the chunks of their kernel-row class: virtual chunk v -> real_cc(v)
    size_t wcur = (size_t)real_cc(0) * a.K * (NMF * 64);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int mf = 0; mf < NMF; ++mf) A0[i][mf] = __builtin_bit_cast(bf4, wpb[i][wcur + mf * 64]);
    stage(real_cc(0));
    swrite_bf(lds);
    __syncthreads();
    int sK = a.K, sDil = a.dil, sStride = a.stride, sPH = a.PH, sBuf = bufsz, sROW = a.ROW, sPacked = a.seg_len;
    asm volatile("" : "+s"(sK), "+s"(sDil), "+s"(sStride), "+s"(sPH), "+s"(sBuf), "+s"(sROW), "+s"(sPacked));
    const int colbase = wn * NT * TM + n_lane;
    int pcb[NT];                                   // bf16 rows are 4-channel groups: no kk * ROW term in the column
#pragma unroll
    for (int j = 0; j < NT; ++j) pcb[j] = pcol[j] - kk * a.ROW;
    int cc = 0, tap = 0, tq = 0, tph = 0;
    auto step_bf = [&](int step, bf4 (&cur)[MT][NMF], bf4 (&nxt)[MT][NMF]) __attribute__((always_inline)) {
      const u32x2* buf = reinterpret_cast<const u32x2*>(lds + (cc & 1) * sBuf);
      if (step + 1 < n_steps) {
        size_t nofs;
        if constexpr (cls_mode) {
          if (tap + 1 == sK) wcur = (size_t)real_cc(cc + 1) * a.K * (NMF * 64);
          else wcur += NMF * 64;
          nofs = wcur;
        } else {
          nofs = (size_t)(step + 1) * (NMF * 64);
        }
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int mf = 0; mf < NMF; ++mf) nxt[i][mf] = __builtin_bit_cast(bf4, wpb[i][nofs + mf * 64]);
      }
      if (tap == 0 && cc + 1 < n_cc) stage(real_cc(cc + 1));
      const int tapoff = tph * sPH + tq;
      bf4 bfr[NMF][NT];
#pragma unroll
      for (int mf = 0; mf < NMF; ++mf)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const int c4 = TM == 32 ? 2 * mf + kk : kk;
          bfr[mf][j] = __builtin_bit_cast(bf4, buf[c4 * sROW + (sPacked ? pcb[j] : colbase + j * TM) + tapoff]);
        }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mf = 0; mf < NMF; ++mf)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[i][j] = MfmaBf<TM>::run(cur[i][mf], bfr[mf][j], acc[i][j]);
      __builtin_amdgcn_sched_barrier(0);
      tq += sDil;
      if (sStride != 1) {
        tq -= sDil;
        tph += sDil;
        if (tph >= sStride) {
          tph -= sStride;
          ++tq;
        }
      }
      if (++tap == sK) {
        tap = 0;
        tq = 0;
        tph = 0;
        if (cc + 1 < n_cc) swrite_bf(lds + ((cc + 1) & 1) * sBuf);
        __syncthreads();
        ++cc;
      }
    };
    int step = 0;
    for (; step + 1 < n_steps; step += 2) {
      step_bf(step, A0, A1);
      step_bf(step + 1, A1, A0);
    }
    if (step < n_steps) step_bf(step, A0, A1);
  } else {
  // two named A-fragment register sets used alternately (no loop-carried copy: with a copy at the end of the tap the
  // compiler waits for the JUST-issued prefetch in the middle of the MFMA phase, one exposed L2 latency per tap)
  // One-tile waves (MT * NT == 1) multiply for only 8 x 64 cycles per step — less than an L2 round trip — and the small
  // problems that pick this shape often run one wave per SIMD: their fragments are requested TWO steps ahead (three sets).
  constexpr int PD = (MT * NT == 1 && !cls_mode) ? 2 : 1;     // prefetch distance in (chunk, tap) steps
  float a0[MT][CPN], a1[MT][CPN], a2[PD == 2 ? MT : 1][CPN];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int cp = 0; cp < CPN; ++cp) a0[i][cp] = wptr[i][wofs + cp * 64];
  if constexpr (PD == 2) {
    if (n_steps > 1) {
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int cp = 0; cp < CPN; ++cp) a1[i][cp] = wptr[i][wstep + cp * 64];
    }
  }

  int* tab = reinterpret_cast<int*>(lds + a.tab_off);
  if (a.tapmajor) {
    // LDS offset of every (k-step, kk): channel row + (phase-de-interleaved) tap offset; padding entries point at 0
    for (int e = tid; e < a.K * CPN * KK; e += RTG_THREADS) {
      const int ks = e / KK, k2 = e - ks * KK;
      const int c = ks / a.TG, j = (ks - c * a.TG) * KK + k2;
      int off = 0;
      if (c < a.Cg && j < a.K_real) {
        const int td = j * a.dil;
        off = c * a.ROW + ((a.stride == 1) ? td : (td % a.stride) * a.PH + td / a.stride);
      }
      tab[e] = off;
    }
  }
  stage(real_cc(0));
  RTG_STAMP(1);
  swrite(lds);
  __syncthreads();
  RTG_STAMP(2);

  int cc = 0, tap = 0;
  // Loop-invariant scalars pinned in SGPRs (the empty asm keeps the compiler from re-loading them from the kernarg
  // segment inside the loop: an s_load + s_waitcnt lgkmcnt(0) per step also drains the LDS reads in flight), the LDS
  // word offset of every k-step row per lane, and the tap offset of strided layers tracked incrementally
  // (tap * dil = q * stride + ph) instead of divided out every step.
  int sK = a.K, sDil = a.dil, sStride = a.stride, sPH = a.PH, sBuf = bufsz, sTapm = a.tapmajor, sPacked = a.seg_len;
  asm volatile("" : "+s"(sK), "+s"(sDil), "+s"(sStride), "+s"(sPH), "+s"(sBuf), "+s"(sTapm), "+s"(sPacked));
  int sRowStep = KK * a.ROW;
  asm volatile("" : "+s"(sRowStep));
  int brow[CPN];
#pragma unroll
  for (int cp = 0; cp < CPN; ++cp) brow[cp] = bbase + cp * KK * a.ROW;
  int tq = 0, tph = 0;                          // tap * dil = tq * stride + tph
  // one (chunk, tap) step: prefetch the next step's A fragments into `nxt`, multiply with `cur`
  auto do_step = [&](int step, float (&cur)[MT][CPN], float (&nxt)[MT][CPN]) __attribute__((always_inline)) {
    const float* buf = lds + (cc & 1) * sBuf;
    if (step + PD < n_steps) {
      // weight offset of the step PD ahead: the next tap of this chunk, or (once per chunk) tap 0 of the next walked chunk
      size_t nofs;
      if constexpr (cls_mode) {
        if (tap + 1 == sK) wofs = (size_t)real_cc(cc + 1) * a.K * wstep;
        else wofs += wstep;
        nofs = wofs;
      } else {
        nofs = (size_t)(step + PD) * wstep;      // all chunks walked in order: a plain running offset
      }
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int cp = 0; cp < CPN; ++cp) nxt[i][cp] = wptr[i][nofs + cp * 64];
    }
    // the next chunk's patch is requested AFTER the weight prefetch, on the chunk's first tap: vmcnt retires in order,
    // so the wait for `nxt` one tap later does not include these loads, the wait two taps later finds them landed
    if (tap == 0 && cc + 1 < n_cc) stage(real_cc(cc + 1));
    const int tapoff = tph * sPH + tq;           // stride 1: tph == 0, PH unused
    const float* bp = buf + tapoff;
    // read phase: all B fragments of this (chunk, tap) into distinct registers, THEN the MFMA phase — the compiler
    // otherwise recycles one register pair and serialises ds_read -> wait -> 2 MFMAs per k-step; with two waves
    // per SIMD one wave's read phase overlaps the other's MFMA phase
    float bf[CPN][NT];
    if (sTapm) {
      int boff[CPN];
#pragma unroll
      for (int cp = 0; cp < CPN; ++cp) boff[cp] = tab[(tap * CPN + cp) * KK + kk];
      if (sPacked) {
#pragma unroll
        for (int cp = 0; cp < CPN; ++cp)
#pragma unroll
          for (int j = 0; j < NT; ++j) bf[cp][j] = buf[boff[cp] + pcol[j] - kk * a.ROW];
      } else {
        const float* b0p = buf + wn * NT * TM + n_lane;
#pragma unroll
        for (int cp = 0; cp < CPN; ++cp)
#pragma unroll
          for (int j = 0; j < NT; ++j) bf[cp][j] = b0p[boff[cp] + j * TM];
      }
    } else if (sPacked) {
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const float* bj = bp + pcol[j];
#pragma unroll
        for (int cp = 0; cp < CPN; ++cp) bf[cp][j] = bj[cp * sRowStep];
      }
    } else {
#pragma unroll
      for (int cp = 0; cp < CPN; ++cp)
#pragma unroll
        for (int j = 0; j < NT; ++j) bf[cp][j] = bp[brow[cp] + j * TM];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int cp = 0; cp < CPN; ++cp)
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = M::run(cur[i][cp], bf[cp][j], acc[i][j]);
    __builtin_amdgcn_sched_barrier(0);
    tq += sDil;                                  // next tap (strided layers have dil == 1: at most one carry)
    if (sStride != 1) {
      tq -= sDil;
      tph += sDil;
      if (tph >= sStride) {
        tph -= sStride;
        ++tq;
      }
    }
    if (++tap == sK) {
      tap = 0;
      tq = 0;
      tph = 0;
      if (cc + 1 < n_cc) swrite(lds + ((cc + 1) & 1) * bufsz);
      __syncthreads();
      ++cc;
    }
  };
  int step = 0;
  if constexpr (PD == 2) {
    for (; step + 2 < n_steps; step += 3) {
      do_step(step, a0, a2);                     // multiply with set k % 3, request step k + 2 into set (k + 2) % 3
      do_step(step + 1, a1, a0);
      do_step(step + 2, a2, a1);
    }
    if (step < n_steps) do_step(step, a0, a2);
    if (step + 1 < n_steps) do_step(step + 1, a1, a0);
  } else {
    for (; step + 1 < n_steps; step += 2) {
      do_step(step, a0, a1);
      do_step(step + 1, a1, a0);
    }
    if (step < n_steps) do_step(step, a0, a1);
  }
  }
  RTG_STAMP(3);

  // ---- epilogue, fast path (plain and polyphase "shuffle" stores): 32-bit element offsets through buffer descriptors,
  // invalid rows / columns loaded from / stored to an out-of-range offset (returned as 0 / dropped by the hardware): no
  // per-element branches, all loads of a tile in flight together, stores issue back to back.  Row m of the GEMM is
  // output channel ch = m / S at phase m % S (S = shuf_S, 1 for a plain conv): element (m, q) lands at position
  // q * S + phase - shuf_P of channel ch; the per-row part (ch, phase: one division per accumulator row) is computed
  // once per m tile, the per-column part once per column tile, an element costs an add, a range check and a select.
  if (a.out_split == 0) {
    const int out_bytes = a.B * a.out_C * a.out_L * 4;
    const rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, out_bytes, 0x00020000);
    const rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)(a.bias ? a.bias : a.out), 0, a.bias ? a.out_C * 4 : 0,
                                                        0x00020000);
    const rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc((void*)(a.mask ? a.mask : a.out), 0, a.mask ? out_bytes : 0,
                                                        0x00020000);
    const rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void*)(a.res ? a.res : a.out), 0, a.res ? out_bytes : 0,
                                                        0x00020000);
    const rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, a.accumulate ? out_bytes : 0, 0x00020000);
    const float mslope = a.mask ? a.mask_slope : 1.f;
    const int S = a.shuf_S;
    const float invS = 1.0f / (float)S;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      if (mt0 + i >= a.n_mt) continue;
      const int mbase = (mt0 + i) * TM;
      float bv[M::NREG];
      unsigned rowoff[M::NREG];          // (ch * out_L + phase - shuf_P) * 4, modulo 2^32 (the column part makes it >= 0)
      int rowph[M::NREG];                // phase - shuf_P; far negative for rows past Mg (fails every range check)
#pragma unroll
      for (int r = 0; r < M::NREG; ++r) {
        const int m = mbase + M::row(lane, r);
        const int mrow = g * a.Mg + m;
        int ch = mrow, ph = 0;
        if (S != 1) {                    // mrow / S through the float reciprocal (mrow < 2^24), one correction step
          ch = (int)((float)mrow * invS);
          ph = mrow - ch * S;
          if (ph < 0) { --ch; ph += S; }
          else if (ph >= S) { ++ch; ph -= S; }
          ph -= a.shuf_P;
        }
        const bool rok = m < a.Mg;
        rowoff[r] = (unsigned)(ch * a.h_n * a.out_L + ph) * 4u;          // h_n == 1 in 1-D
        rowph[r] = rok ? ph : -(1 << 28);
        bv[r] = buf_load(rb, rok ? (unsigned)ch * 4u : RTG_OOB);
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        int q = q_blk + q0 + (wn * NT + j) * TM + n_lane;
        int b = b0;
        bool ok = true;
        int seg = 0;
        if (packed) {
          seg = q / a.Q;
          q -= seg * a.Q;
          b = b0 + seg;
          ok = seg < a.seg_nb && (cls_mode ? seg < rc.cnt : b < a.B);
        }
        ok = ok && q < a.Q;
        int hh = 0;
        if (a.two_d) {                     // clip -> (batch item, output row)
          int item = b / a.h_n;
          hh = b - item * a.h_n;
          if (cls_mode) {
            item = rc.item;
            hh = rc.first + (rc.j0 + seg) * a.h_stride;
          }
          b = item;
        }
        const int qs = ok ? q * S : -(1 << 28);
        const unsigned col = ((unsigned)(b * a.out_C * a.h_n + hh) * (unsigned)a.out_L + (unsigned)(q * S)) * 4u;
        // eight accumulator rows at a time: their operand loads are in flight together, and the epilogue's registers
        // (offsets + three optional operands) stay below the main loop's — the block shape's occupancy is set there
        constexpr int RH = M::NREG > 8 ? 8 : M::NREG;
#pragma unroll
        for (int r0 = 0; r0 < M::NREG; r0 += RH) {
          unsigned off[RH];
          float mv[RH], rv[RH], av[RH];
#pragma unroll
          for (int r = 0; r < RH; ++r)
            off[r] = ((unsigned)(qs + rowph[r0 + r]) < (unsigned)a.out_L) ? col + rowoff[r0 + r] : RTG_OOB;
          // optional operands: uniform branches per tile (an absent operand costs no load instructions at all; a load
          // through a zero-record descriptor would still occupy the address unit for a full wave)
          if (a.mask) {
#pragma unroll
            for (int r = 0; r < RH; ++r) mv[r] = buf_load(rm, off[r]);
          } else {
#pragma unroll
            for (int r = 0; r < RH; ++r) mv[r] = 1.f;
          }
          if (a.res) {
#pragma unroll
            for (int r = 0; r < RH; ++r) rv[r] = buf_load(rr, off[r]);
          } else {
#pragma unroll
            for (int r = 0; r < RH; ++r) rv[r] = 0.f;
          }
          if (a.accumulate) {
#pragma unroll
            for (int r = 0; r < RH; ++r) av[r] = buf_load(ra, off[r]);
          } else {
#pragma unroll
            for (int r = 0; r < RH; ++r) av[r] = 0.f;
          }
#pragma unroll
          for (int r = 0; r < RH; ++r) {
            float v = acc[i][j][r0 + r] + bv[r0 + r];
            // (one explicit fused multiply-add: every kernel that serves a layer rounds this step the same way,
            // rtg_resconv.hip included, whatever the compiler would contract)
            v = __builtin_fmaf(v, mv[r] > 0.f ? 1.f : mslope, rv[r]) * a.out_scale;
            if (a.act == RTG_ACT_LRELU) v = rtg_lrelu(v, a.act_slope);
            else if (a.act == RTG_ACT_TANH) v = tanhf(v);
            v += av[r];
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ro, off[r], 0, 0);
          }
        }
      }
    }
    RTG_STAMP(4);
    return;
  }

  // ---- epilogue, general path (concat-split store)
#ifdef RTG_EXP_NOGENERAL
  return;                                 // ablation: code-size experiment (results of the general path are dropped)
#endif
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    if (mt0 + i >= a.n_mt) continue;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      int q = q_blk + q0 + (wn * NT + j) * TM + n_lane;
      int b = b0;
      int seg = 0;
      if (packed) {
        seg = q / a.Q;
        q -= seg * a.Q;
        b = b0 + seg;
        if (seg >= a.seg_nb || (cls_mode ? seg >= rc.cnt : b >= a.B)) continue;
      }
      if (q >= a.Q) continue;
      int hh = 0;
      if (a.two_d) {                     // clip -> (batch item, output row)
        int item = b / a.h_n;
        hh = b - item * a.h_n;
        if (cls_mode) {
          item = rc.item;
          hh = rc.first + (rc.j0 + seg) * a.h_stride;
        }
        b = item;
      }
#pragma unroll
      for (int r = 0; r < M::NREG; ++r) {
        const int m = (mt0 + i) * TM + M::row(lane, r);
        if (m >= a.Mg) continue;
        const int mrow = g * a.Mg + m;
        int ch = mrow, u = q;
        if (a.shuf_S > 1) {
          ch = mrow / a.shuf_S;
          u = q * a.shuf_S + (mrow - ch * a.shuf_S) - a.shuf_P;
          if (u < 0 || u >= a.out_L) continue;
        }
        float* dst = a.out;
        size_t idx;
        if (a.out_split > 0) {
          if (ch >= a.out_split) {
            dst = a.out2;
            idx = ((size_t)b * (a.out_C - a.out_split) + (ch - a.out_split)) * a.out_L + u;
          } else {
            idx = ((size_t)b * a.out_split + ch) * a.out_L + u;
          }
          if (!dst) continue;
        } else {
          idx = (((size_t)b * a.out_C + ch) * a.h_n + hh) * a.out_L + u;   // h_n == 1, hh == 0 in 1-D
        }
        float v = acc[i][j][r];
        if (a.bias) v += a.bias[ch];
        if (a.mask) v *= (a.mask[idx] > 0.f ? 1.f : a.mask_slope);
        if (a.res) v += a.res[idx];
        v *= a.out_scale;
        if (a.act == RTG_ACT_LRELU) v = rtg_lrelu(v, a.act_slope);
        else if (a.act == RTG_ACT_TANH) v = tanhf(v);
        if (a.accumulate) v += dst[idx];
        dst[idx] = v;
      }
    }
  }
  RTG_STAMP(4);
}

template <int TM, int MT, int NT, int MAXIT, bool CLS, bool BF>
__global__ __launch_bounds__(RTG_THREADS) void conv1d_mfma_group_kernel(const GroupArgs ga) {
  int pid = 0;
  unsigned start = 0;
  for (int i = 0; i + 1 < ga.n; ++i)
    if (blockIdx.x >= ga.blk_end[i]) {
      pid = i + 1;
      start = ga.blk_end[i];
    }
  conv1d_mfma_body<TM, MT, NT, MAXIT, CLS, BF>(ga.p[pid], blockIdx.x - start);     // every grid is a multiple of 8 blocks
}

template <int TM, int MT, int NT, int MAXIT, bool CLS, bool BF>
int launch_group_cls(const GroupArgs& ga, size_t lds_bytes, hipStream_t s) {
  auto k = conv1d_mfma_group_kernel<TM, MT, NT, MAXIT, CLS, BF>;
  if (lds_bytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return -(1000 + (int)e);
  }
  RTG_KLAUNCH(k, dim3(ga.blk_end[ga.n - 1]), dim3(RTG_THREADS), lds_bytes, s, ga);
  return rtg_launch_status();
}

template <int TM, int MT, int NT, int MAXIT>
int launch_group_it(const GroupArgs& ga, size_t lds_bytes, int bf, hipStream_t s) {
  bool cls = false;
  for (int i = 0; i < ga.n; ++i) {
    const bool c = ga.p[i].two_d && ga.p[i].h_mode == 1 && ga.p[i].h_stride > 1;
    if (i > 0 && c != cls) return RTG_EINVAL;      // one kernel instance per launch
    cls = c;
  }
  if (bf) {
    for (int i = 0; i < ga.n; ++i)
      if (ga.p[i].tapmajor) return RTG_EINVAL;
    return cls ? launch_group_cls<TM, MT, NT, MAXIT, true, true>(ga, lds_bytes, s)
               : launch_group_cls<TM, MT, NT, MAXIT, false, true>(ga, lds_bytes, s);
  }
  return cls ? launch_group_cls<TM, MT, NT, MAXIT, true, false>(ga, lds_bytes, s)
             : launch_group_cls<TM, MT, NT, MAXIT, false, false>(ga, lds_bytes, s);
}

template <int TM, int MT, int NT>
int launch_group(const GroupArgs& ga, size_t lds_bytes, int bf, hipStream_t s) {
  int pw = 0;
  for (int i = 0; i < ga.n; ++i) pw = ga.p[i].PW > pw ? ga.p[i].PW : pw;
  if (pw <= 3 * 64) return launch_group_it<TM, MT, NT, 3>(ga, lds_bytes, bf, s);
  if (pw <= 5 * 64) return launch_group_it<TM, MT, NT, 5>(ga, lds_bytes, bf, s);
  return launch_group_it<TM, MT, NT, RTG_PW_MAX / 64>(ga, lds_bytes, bf, s);
}

}  // namespace rtg_cv

#define RTG_CONV_DEFINE(tm, mt, nt)                                                                         \
  int rtg_conv1d_launch_group_##tm##_##mt##_##nt(const rtg_cv::GroupArgs& ga, size_t lds_bytes, int bf,        \
                                                 hipStream_t s) {                                              \
    return rtg_cv::launch_group<tm, mt, nt>(ga, lds_bytes, bf, s);                                           \
  }
