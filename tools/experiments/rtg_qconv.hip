// NOT PART OF librtg.so.  Round-6 experiment, kept with its measurements (profiles/r06_qconv_experiment.txt, DESIGN.md section 3,
// round 6): bit-identical to the library's block shapes, a tie on the large stride-1 layers (126 vs 127-130 TFLOP/s), slower on
// short rows and on the 2-tap / stride-3 operators (64-position column granularity).  To try it again: copy it into
// transtacos-retunegan_amd/csrc/, declare rtg_qconv_candidates / rtg_qconv_launch in rtg_dconv.hip, put the candidates in front of
// rtg_dconv_candidates' list and route shape digits >= 6 of rtg_dconv_launch to rtg_qconv_launch (git show cc2bb1c~3 has the hooks).
//
// rtg_qconv.hip — "quad-column" form of the dense-layer conv kernel (round 6): the same layers as rtg_dconv.hip in 1-D —
//   DiscriminatorP convs.1-4 ((5,1) kernels, stride 3 / 3 / 3 / 1) and DiscriminatorS convs.5 (k5)   discrminator.py:44,155-163
// forward (5 taps at stride 1 / 3), backward-data of the stride-1 layers (5 taps) and the polyphase backward-data of the
// stride-3 layers (2 taps, rows = (channel, phase), interleaving store) — on the SAME 16-byte-fragment weight image
// (RtgPackJob.frag16), with the same accumulation order per output element (chunk, tap, channel group, one fused multiply-add
// each): bit-identical to every other block shape of a layer.
//
// What rtg_dconv_kernel.h still pays per matrix instruction, and what changes here.  There a lane's four k-steps of a chunk
// come from ONE 16-byte LDS read of a patch stored [kgrp][position][4 channels] — so the input has to be TRANSPOSED on its way
// into LDS: every staged element is a 4-byte global load, an activation and a quarter of a 16-byte LDS write, and a chunk of
// 16 channels feeds K x 4 matrix instructions per 16-column tile before the next barrier.  The 2-tap polyphase instances
// (backward-data of the stride-3 layers: 2.9 ms of the config-2 step) sit at 0.44 matrix-pipe occupancy on that: a staging
// round and a barrier per 2 taps (rounds 4-5: two chunks per barrier, a second register set, longer
// request distances — all measured neutral; the per-element staging work itself was never removed).
// Here the four matrix instructions that share a fetch differ in their COLUMN, not in their k-step:
//   columns   lane (n16, kgrp) of column tile ct owns the QUAD of four consecutive output positions 4 qic .. 4 qic + 3 of one
//             clip; instruction j of a group computes column j of every lane's quad into its own accumulator tile, so a lane
//             ends up with four consecutive positions of four rows: the epilogue is 16-byte loads and stores where the row
//             length allows;
//   patch     RAW rows, [channel][virtual position] exactly as they lie in the NCW tensor: staged with unaligned 16-byte global
//             loads (one per four positions of a channel, full rate at any 4-byte alignment, profiles/r05_unaligned_16B_loads.txt),
//             row ends zeroed by a per-slot element mask, activation applied to the four values, ONE aligned ds_write_b128 —
//             no transposition, a quarter of the load / write instructions and none of the per-position index arithmetic;
//   windows   for channel group kq a lane reads its quad's input window — the aligned quads 4 S qic .. (2 quads at stride 1,
//             4 at stride 3) of channel row 4 kq + kgrp — once per chunk; tap t of column j is element S j + t of that window: a
//             register NAME, no LDS read per tap.  20 (K = 5) or 8 (K = 2) matrix instructions per pair of 16-byte reads where
//             the transposed patch gives 4 per read;
//   weights   the frag16 image as it is: a lane's 16 bytes at (chunk, tap) are the four channel groups kq;
//   barriers  NCB chunks (32 channels) per barrier; the patch of the next step is requested a step ahead and written behind
//             the step's matrix instructions.
// A clip's segment is S (QPC - 1) + NRQ quads of LDS (QPC = ceil(Q / 4) output quads per clip): the columns of a block are a
// window of the dense (clip, quad) sequence, a row whose length is no multiple of 4 wastes its last quad's tail
// (rows of 34 / 21 / 15 / 10 positions: 6 / 12 / 6 / 17 % of the columns).
// Exposed as block-shape codes 8000 + 100 * (6 + shape) + NT (RtgConv1dDesc.tile_cfg) in rtg_dconv_candidates' list; the tuner
// times them against the other shapes.
#include <type_traits>

#include "rtg_common.h"

namespace rtg_qc {

using rsrc_t = __amdgpu_buffer_rsrc_t;
using u32x4 = unsigned __attribute__((ext_vector_type(4)));
#define QC_OOB 0x80000000u

struct QArgs {
  const float *x, *wp, *bias, *mask, *res;
  float* out;
  int B, C, L_in, Mg, n_cc, Q, pad, out_C, out_L, shuf_S, shuf_P;
  int pre, act, accumulate;
  float pre_slope, mask_slope, out_scale, act_slope;
  int QPC, PQ, NQ;             // output quads per clip, LDS quads per clip segment, B * QPC
  int WQ, pitch;               // staged quads per channel row (widest block), floats between channel rows (multiple of 64)
  int n_mb, total, per_xcd;    // row blocks, work items, items per XCD
  int x_bytes, out_bytes, mask_bytes, res_bytes;
  int vec;                     // 1: rows are whole quads and contiguous (no shuffle): 16-byte epilogue accesses
};

__device__ __forceinline__ f32x4 qc_load4(rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ float qc_load(rsrc_t r, unsigned voff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, 0, 0));
}

// quads a lane reads per (channel, column tile): the window of outputs 4 qic .. 4 qic + 3 is 3 S + K input positions
constexpr int nrq(int S, int K) { return (3 * S + K + 3) / 4; }
// the widest window (in quads) a block of NT column tiles stages, rows of at least 5 positions (QPC >= 2): S quads per output
// quad, NRQ - S more per clip segment it touches
constexpr int max_wq(int NT, int S, int K) { return NT * 16 * S + (NT * 16 / 2 + 1) * (nrq(S, K) - S > 0 ? nrq(S, K) - S : 0) + nrq(S, K); }
constexpr int max_slots(int NT, int S, int K, int NCB, int WB) { return (NCB * 16 * max_wq(NT, S, K) + WB * 64 - 1) / (WB * 64); }

// RW16: 16-row tiles per wave; WB: waves per block (stacked along the rows); NT: column tiles (16 quads = 64 positions) per
// block (= per wave); S: stride of the walk; K: taps; NCB: 16-channel chunks per barrier
template <int RW16, int WB, int NT, int S, int K, int NCB>
__global__ __launch_bounds__(WB * 64, 2) void qconv_kernel(const QArgs a) {
  constexpr int NRQ = nrq(S, K);
  constexpr int NTHR = WB * 64;
  constexpr int MAXSL = max_slots(NT, S, K, NCB, WB);
  // two window register sets (the next chunk's windows are read during this chunk's matrix instructions) where they fit
  constexpr bool WIN2 = NT * NRQ <= 2;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n16 = lane & 15, kgrp = lane >> 4;
  const int item = (int)(blockIdx.x & 7u) * a.per_xcd + (int)(blockIdx.x >> 3);
  if (item >= a.total) return;
  const int mb = item % a.n_mb, nt = item / a.n_mb;
  const int bq0 = nt * (NT * 16);                    // first quad of the block in the dense (clip, quad) sequence
  const int clip0 = bq0 / a.QPC, qic0 = bq0 - clip0 * a.QPC;
  const int wq0 = S * qic0;                          // segment quad of clip0 that sits at LDS quad 0
  const int bufF = NCB * 16 * a.pitch;               // floats per LDS buffer

  // ---- staging slots: slot k of this thread is quad w of channel row ch_l (of the step's NCB * 16 channels), fixed for the
  // whole walk; per slot the element offset in x without the step's channel base, the validity of its four positions, the
  // LDS destination
  int sl_eoff[MAXSL], sl_dst[MAXSL];
  unsigned sl_msk[MAXSL];
#pragma unroll
  for (int k = 0; k < MAXSL; ++k) {
    const int idx = tid + k * NTHR;
    const int ch_l = idx / a.WQ, w = idx - ch_l * a.WQ;
    const bool live = ch_l < NCB * 16;
    const int g = w + wq0;
    const int seg = g / a.PQ, u = g - seg * a.PQ;
    const int clip = clip0 + seg;
    const int p0 = 4 * u - a.pad;                    // input position of the quad's element 0
    const bool ok = live && clip < a.B;
    unsigned m = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) m |= (ok && (unsigned)(p0 + e) < (unsigned)a.L_in) ? (1u << e) : 0u;
    sl_msk[k] = m | (live ? 16u : 0u);
    sl_eoff[k] = (clip * a.C + ch_l) * a.L_in + p0;
    sl_dst[k] = ch_l * a.pitch + 4 * w;
    if (!ok || m == 0) sl_eoff[k] = -(1 << 30);      // nothing to read: out of range whatever the step (a quad of zeros)
  }
  const rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
  const int step_elems = NCB * 16 * a.L_in;          // elements between the channel bases of consecutive steps
  const float wslope = a.pre ? a.pre_slope : 1.f;
  const int n_steps = a.n_cc / NCB;                  // (host: n_cc is a multiple of NCB)
  f32x4 st[MAXSL];
  // (unconditional: past the last step the offsets are out of range and the loads return zeros nobody writes)
  auto stage_issue = [&](int s) __attribute__((always_inline)) {
    const int base = s < n_steps ? s * step_elems : -(1 << 30);
#pragma unroll
    for (int k = 0; k < MAXSL; ++k) {
      const int e = sl_eoff[k] + base;
      // the very first elements of the tensor cannot be addressed from before its start (a left pad of clip 0, channel 0):
      // that one quad loads from element 0 and is shifted when written
      st[k] = qc_load4(rx, e >= 0 ? (unsigned)e * 4u : (e > -8 && s < n_steps ? 0u : QC_OOB), 0);
    }
  };
  auto stage_write = [&](int s, float* buf) __attribute__((always_inline)) {
    const int base = s * step_elems;
#pragma unroll
    for (int k = 0; k < MAXSL; ++k) {
      f32x4 v = st[k];
      asm volatile("" : "+v"(v));                    // keep the consumption (and its wait) here, below the multiplications
      const int e = sl_eoff[k] + base;
      if (e < 0 && e > -8) {                         // (one thread of the grid, first step only)
        const int sh = -e;
        f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int jj = 0; jj < 4; ++jj)
            if (i - jj == sh) t[i] = v[jj];
        v = t;
      }
      const unsigned m = sl_msk[k];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float t = (m >> i) & 1u ? v[i] : 0.f;
        v[i] = t > 0.f ? t : t * wslope;
      }
      if (m & 16u) *reinterpret_cast<f32x4*>(buf + sl_dst[k]) = v;
    }
  };

  // ---- operand addressing: this lane's quad of column tile ct -> float offset of its window in a channel row
  int woff[NT];
#pragma unroll
  for (int ct = 0; ct < NT; ++ct) {
    int qi = bq0 + ct * 16 + n16;
    if (qi > a.NQ - 1) qi = a.NQ - 1;                // junk column: a staged quad, dropped in the epilogue
    const int clip = qi / a.QPC, qic = qi - clip * a.QPC;
    woff[ct] = 4 * ((clip - clip0) * a.PQ + S * qic - wq0) + kgrp * a.pitch;
  }
  const int n_mt16 = (a.Mg + 15) >> 4;
  const f32x4* aptr[RW16];
#pragma unroll
  for (int i = 0; i < RW16; ++i) {
    int mt = (mb * WB + wave) * RW16 + i;
    if (mt > n_mt16 - 1) mt = n_mt16 - 1;            // clamped duplicate tile, dropped in the epilogue
    aptr[i] = reinterpret_cast<const f32x4*>(a.wp) + (size_t)mt * a.n_cc * K * 64 + lane;
  }

  f32x4 acc[RW16][NT][4];
#pragma unroll
  for (int i = 0; i < RW16; ++i)
#pragma unroll
    for (int ct = 0; ct < NT; ++ct)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][ct][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // weight fragments of a chunk: K x RW16 coalesced 1-KB loads from L2, requested a chunk ahead into the other set
  struct AFr {
    f32x4 a[K][RW16];
  };
  auto fetch_a = [&](AFr& f, int rc) __attribute__((always_inline)) {
    const int c = rc < a.n_cc ? rc : a.n_cc - 1;     // (past the end: re-read the last chunk, never used)
#pragma unroll
    for (int t = 0; t < K; ++t)
#pragma unroll
      for (int i = 0; i < RW16; ++i) f.a[t][i] = aptr[i][(size_t)(c * K + t) * 64];
  };
  // input windows of a chunk: for channel group kq the aligned quads of row 4 kq + kgrp this lane's outputs read
  struct Win {
    f32x4 q[4][NT][NRQ];
  };
  auto fetch_w = [&](Win& w, const float* buf, int cc) __attribute__((always_inline)) {
#pragma unroll
    for (int kq = 0; kq < 4; ++kq)
#pragma unroll
      for (int ct = 0; ct < NT; ++ct)
#pragma unroll
        for (int r = 0; r < NRQ; ++r)
          w.q[kq][ct][r] = *reinterpret_cast<const f32x4*>(buf + (cc * 16 + 4 * kq) * a.pitch + woff[ct] + 4 * r);
  };
  // (inline asm with the accumulator tied to the destination, as in rtg_dconv_kernel.h)
  auto mma = [&](const AFr& f, const Win& w) __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < K; ++t)
#pragma unroll
      for (int kq = 0; kq < 4; ++kq)
#pragma unroll
        for (int i = 0; i < RW16; ++i)
#pragma unroll
          for (int ct = 0; ct < NT; ++ct)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int e = S * j + t;               // element of the window: a register name
              const float av = f.a[t][i][kq], bv = w.q[kq][ct][e >> 2][e & 3];
              asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i][ct][j]) : "v"(av), "v"(bv));
            }
  };

  // ---- prologue: step 0 staged and published, step 1 requested, the first chunk's weights requested
  stage_issue(0);
  stage_write(0, lds);
  __syncthreads();
  stage_issue(1);
  AFr fa0, fa1;
  fetch_a(fa0, 0);
  Win w0;
  [[maybe_unused]] Win w1;

  for (int s = 0; s < n_steps; ++s) {
    const float* buf = lds + (s & 1) * bufF;
    float* bufn = lds + ((s + 1) & 1) * bufF;
    fetch_w(w0, buf, 0);
#pragma unroll
    for (int cc = 0; cc < NCB; ++cc) {
      AFr& fc = (cc & 1) ? fa1 : fa0;
      AFr& fn = (cc & 1) ? fa0 : fa1;
      fetch_a(fn, s * NCB + cc + 1);
      if constexpr (WIN2) {
        Win& wc = (cc & 1) ? w1 : w0;
        Win& wn = (cc & 1) ? w0 : w1;
        if (cc + 1 < NCB) fetch_w(wn, buf, cc + 1);
        __builtin_amdgcn_sched_barrier(0);
        mma(fc, wc);
      } else {
        __builtin_amdgcn_sched_barrier(0);
        mma(fc, w0);
        if (cc + 1 < NCB) {
          __builtin_amdgcn_sched_barrier(0);
          fetch_w(w0, buf, cc + 1);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    static_assert((NCB & 1) == 0, "the weight-fragment sets alternate per chunk: an even count per step keeps them in place");
    // publish the next step's patch: its buffer was last read before the previous barrier
    if (s + 1 < n_steps) stage_write(s + 1, bufn);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    stage_issue(s + 2);
  }

  // (the matrix instructions are inline asm: the compiler does not know their results are still in flight)
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  // ---- epilogue: out = act(((acc + bias) * dmask + res) * out_scale) (+ out): the arithmetic and rounding of the general
  // kernel (rtg_conv1d_kernel.h) and of rtg_dconv_kernel.h.  Row m' of the GEMM is output channel m' / S_out at phase
  // m' % S_out (polyphase backward-data: interleaving 4-byte stores); without it and with rows of whole quads a lane's four
  // positions are ONE 16-byte access
  const rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, a.out_bytes, 0x00020000);
  const rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)(a.bias ? a.bias : a.out), 0, a.bias ? a.out_C * 4 : 0, 0x00020000);
  const rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc((void*)(a.mask ? a.mask : a.out), 0, a.mask ? a.mask_bytes : 0, 0x00020000);
  const rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void*)(a.res ? a.res : a.out), 0, a.res ? a.res_bytes : 0, 0x00020000);
  const float mslope = a.mask ? a.mask_slope : 1.f;
  const int So = a.shuf_S;
  const float invS = 1.0f / (float)So;
  auto finish = [&](float accv, float bv, float mv, float rv, float av) __attribute__((always_inline)) {
    float v = accv + bv;
    v = __builtin_fmaf(v, mv > 0.f ? 1.f : mslope, rv) * a.out_scale;
    if (a.act == RTG_ACT_LRELU) v = rtg_lrelu(v, a.act_slope);
    else if (a.act == RTG_ACT_TANH) v = tanhf(v);
    return v + av;
  };
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
      rowoff[r] = (unsigned)(ch * a.out_L + ph);
      rowph[r] = rok ? ph : -(1 << 28);
      bv[r] = qc_load(rb, rok ? (unsigned)ch * 4u : QC_OOB);
    }
#pragma unroll
    for (int ct = 0; ct < NT; ++ct) {
      const int qi = bq0 + ct * 16 + n16;
      const int clip = qi / a.QPC, qic = qi - clip * a.QPC;
      const bool qv = qi < a.NQ;
      const unsigned col0 = (unsigned)(clip * a.out_C) * (unsigned)a.out_L;
      if (a.vec) {
        // rows of whole quads, no interleave: element offsets of the four positions are consecutive and 16-byte aligned
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool ok = qv && rowph[r] >= 0;
          const unsigned off = ok ? (col0 + rowoff[r] + 4u * (unsigned)qic) * 4u : QC_OOB;
          const f32x4 one = {1.f, 1.f, 1.f, 1.f}, zero = {0.f, 0.f, 0.f, 0.f};
          const f32x4 mv = a.mask ? qc_load4(rm, off, 0) : one;
          const f32x4 rv = a.res ? qc_load4(rr, off, 0) : zero;
          const f32x4 av = a.accumulate ? qc_load4(ro, off, 0) : zero;
          f32x4 v;
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = finish(acc[i][ct][j][r], bv[r], mv[j], rv[j], av[j]);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ro, off, 0, 0);
        }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int q = 4 * qic + j;
          const int qs = (qv && q < a.Q) ? q * So : -(1 << 28);
          const unsigned col = col0 + (unsigned)(q * So);
          unsigned off[4];
          bool ok[4];
          float mv[4], rv[4], av[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            ok[r] = (unsigned)(qs + rowph[r]) < (unsigned)a.out_L;
            off[r] = ok[r] ? (col + rowoff[r]) * 4u : QC_OOB;
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) mv[r] = a.mask ? qc_load(rm, off[r]) : 1.f;
#pragma unroll
          for (int r = 0; r < 4; ++r) rv[r] = a.res ? qc_load(rr, off[r]) : 0.f;
#pragma unroll
          for (int r = 0; r < 4; ++r) av[r] = a.accumulate ? qc_load(ro, off[r]) : 0.f;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float v = finish(acc[i][ct][j][r], bv[r], mv[r], rv[r], av[r]);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ro, off[r], 0, 0);
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------- host
struct QShape {
  int rw16, wb;
};
// code digit 6, 7, 8: rows per block 256, 128, 128
constexpr QShape kQShapes[] = {{2, 8}, {1, 8}, {2, 4}};
constexpr int kNumQShapes = 3;
constexpr int kNCB = 2;

struct QGeo {
  int QPC, PQ, NQ, WQ, pitch, n_mb, n_ct;
  size_t lds_bytes;
};

// geometry of shape `si` with NT column tiles per block for descriptor d; false: the shape does not serve it
inline bool geometry(const RtgConv1dDesc* d, int si, int NT, QGeo* g) {
  if (si < 0 || si >= kNumQShapes || (NT != 1 && NT != 2)) return false;
  const int S = d->stride, K = d->K;
  const int NRQ = nrq(S, K);
  const int rows = kQShapes[si].rw16 * kQShapes[si].wb * 16;
  if (d->Mg < rows / 2) return false;                                 // (mostly clamped duplicate tiles)
  g->QPC = (d->Q + 3) / 4;
  if (g->QPC < 2) return false;                                       // rows of at least 5 positions (max_wq)
  g->PQ = S * (g->QPC - 1) + NRQ;
  // the segment must hold every input position a clip's outputs read: (Q - 1) S + K positions from virtual position 0
  if (4 * g->PQ < (d->Q - 1) * S + K) return false;
  g->NQ = d->B * g->QPC;
  g->n_ct = rtg_ceil_div(g->NQ, NT * 16);
  g->n_mb = rtg_ceil_div(rtg_ceil_div(d->Mg, 16), kQShapes[si].rw16 * kQShapes[si].wb);
  // widest staged window over the blocks: last needed quad - first + 1 (the blocks differ in how many clip boundaries they hold)
  int wq = 0;
  for (int t = 0; t < g->n_ct; ++t) {
    const int bq0 = t * NT * 16;
    int bq1 = bq0 + NT * 16 - 1;
    if (bq1 > g->NQ - 1) bq1 = g->NQ - 1;
    const int c0 = bq0 / g->QPC, q0 = bq0 - c0 * g->QPC, c1 = bq1 / g->QPC, q1 = bq1 - c1 * g->QPC;
    const int need = (c1 - c0) * g->PQ + S * q1 - S * q0 + NRQ;
    wq = need > wq ? need : wq;
    if (g->QPC >= NT * 16 && t > g->QPC / (NT * 16) + 2) break;    // long rows: every later block repeats one of these
  }
  g->WQ = wq;
  if (wq > max_wq(NT, S, K)) return false;
  g->pitch = (4 * wq + 63) / 64 * 64;
  g->lds_bytes = (size_t)2 * kNCB * 16 * g->pitch * sizeof(float);
  return g->lds_bytes <= 150 * 1024;
}

// which (block shape, column tiles, stride) instances exist — the candidate list and the dispatch follow this one rule (no
// instance may spill: build.py).  Left out: 32 rows per wave with two column tiles (64 accumulators next to the windows), two
// column tiles of 4-quad windows at stride 3 (128 window registers), the 4-wave block at stride 3 (8 staging slots per thread)
constexpr bool qc_built(int rw16, int wb, int nt, int S) {
  return !(rw16 == 2 && nt == 2) && !(nt == 2 && S == 3) && !(wb == 4 && S == 3);
}

template <int RW16, int WB, int NT, int S, int K>
int launch(const QArgs& a, size_t lds_bytes, hipStream_t s) {
  auto k = qconv_kernel<RW16, WB, NT, S, K, kNCB>;
  static std::atomic<unsigned> optin{0};              // (> 64 KB of dynamic LDS: opt-in per kernel and device)
  if (lds_bytes > 64 * 1024 && rtg_lds_optin((const void*)k, optin) != RTG_OK) return RTG_ERANGE;
  RTG_KLAUNCH(k, dim3((unsigned)(8 * a.per_xcd)), dim3(WB * 64), lds_bytes, s, a);
  return rtg_launch_status();
}

template <int RW16, int WB, int NT>
int launch_sk(const QArgs& a, int S, int K, size_t lds_bytes, hipStream_t s) {
  if (S == 1 && K == 5) return launch<RW16, WB, NT, 1, 5>(a, lds_bytes, s);
  if (S == 1 && K == 2) return launch<RW16, WB, NT, 1, 2>(a, lds_bytes, s);
  if constexpr (qc_built(RW16, WB, NT, 3)) {
    if (S == 3 && K == 5) return launch<RW16, WB, NT, 3, 5>(a, lds_bytes, s);
  }
  return RTG_EINVAL;
}

}  // namespace rtg_qc

using namespace rtg_qc;

// does a quad-column shape serve descriptor d at all (the caller has checked rtg_dconv.hip's eligibility: dense 1-D / 2-D
// layer with the fragment image)?  1-D rows, fp32 operands and tensors, the three (stride, taps) walks, whole steps of chunks
static bool qconv_eligible(const RtgConv1dDesc* d) {
  if (d->h_k > 1 || d->h_n > 1 || d->bf16 || d->io_bf16 || d->dil != 1 || d->groups != 1) return false;
  if (!((d->stride == 1 && (d->K == 5 || d->K == 2)) || (d->stride == 3 && d->K == 5))) return false;
  if (d->Cg % (16 * kNCB) != 0 || d->Cg < 32 || d->Q < 5 || d->pad < 0 || d->pad > 4) return false;
  if ((long long)d->B * ((d->Q + 3) / 4) >= (1ll << 26)) return false;
  return true;
}

// block-shape codes 8000 + 100 * (6 + shape) + NT that serve d, best guess first (called by rtg_dconv_candidates)
int rtg_qconv_candidates(const RtgConv1dDesc* d, int* codes, int max) {
  if (!qconv_eligible(d)) return 0;
  struct Cand {
    int code;
    double score;
  } c[2 * kNumQShapes];
  int n = 0;
  for (int si = 0; si < kNumQShapes; ++si)
    for (int NT = 1; NT <= 2; ++NT) {
      QGeo g;
      if (!geometry(d, si, NT, &g)) continue;
      if (!qc_built(kQShapes[si].rw16, kQShapes[si].wb, NT, d->stride)) continue;
      const long long blocks = (long long)g.n_mb * g.n_ct;
      const double slots = 256.0 * (kQShapes[si].wb == 4 ? 2 : 1);
      const double rounds = (double)blocks / slots;
      const double eff = rounds / (double)(long long)(rounds + 0.999999);
      const int rows = kQShapes[si].rw16 * kQShapes[si].wb * 16;
      const double rows_eff = (double)d->Mg / (double)(g.n_mb * rows);
      c[n].code = 8000 + 100 * (6 + si) + NT;
      c[n].score = eff * rows_eff;
      ++n;
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

// wp: the 16-byte-fragment image (the caller has skipped the standard image in front of it)
int rtg_qconv_launch(const RtgConv1dDesc* d, int si, int NT, const float* x, const float* wp16, const float* bias,
                     const float* mask, const float* res, float* out, hipStream_t s) {
  if (!qconv_eligible(d)) return RTG_EINVAL;
  QGeo g;
  if (!geometry(d, si, NT, &g)) return RTG_EINVAL;
  if (!qc_built(kQShapes[si].rw16, kQShapes[si].wb, NT, d->stride)) return RTG_EINVAL;
  QArgs a;
  a.x = x; a.wp = wp16; a.bias = bias; a.mask = mask; a.res = res; a.out = out;
  a.B = d->B; a.C = d->C1; a.L_in = d->L_in; a.Mg = d->Mg; a.n_cc = d->Cg / 16; a.Q = d->Q; a.pad = d->pad;
  a.out_C = d->out_C; a.out_L = d->out_L; a.shuf_S = d->shuf_S; a.shuf_P = d->shuf_P;
  a.pre = d->pre_mode == RTG_PRE_LRELU ? 1 : 0; a.act = d->act; a.accumulate = d->accumulate;
  a.pre_slope = d->pre_slope; a.mask_slope = d->mask_slope; a.out_scale = d->out_scale; a.act_slope = d->act_slope;
  a.QPC = g.QPC; a.PQ = g.PQ; a.NQ = g.NQ; a.WQ = g.WQ; a.pitch = g.pitch;
  a.n_mb = g.n_mb;
  const long long total = (long long)g.n_mb * g.n_ct;
  if (total > (1ll << 28)) return RTG_ERANGE;
  a.total = (int)total;
  a.per_xcd = rtg_ceil_div(total, 8);
  a.x_bytes = d->B * d->C1 * d->L_in * 4;
  const int out_elems = d->B * d->out_C * d->out_L;
  a.out_bytes = a.mask_bytes = a.res_bytes = out_elems * 4;
  a.vec = (d->shuf_S == 1 && d->Q == d->out_L && (d->Q & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0 &&
           (!mask || (reinterpret_cast<uintptr_t>(mask) & 15) == 0) && (!res || (reinterpret_cast<uintptr_t>(res) & 15) == 0)) ? 1 : 0;
#define RTG_QC(S_, N_) \
  if (si == S_ && NT == N_) return launch_sk<kQShapes[S_].rw16, kQShapes[S_].wb, N_>(a, d->stride, d->K, g.lds_bytes, s);
  RTG_QC(0, 1) RTG_QC(1, 1) RTG_QC(1, 2) RTG_QC(2, 1)
#undef RTG_QC
  return RTG_EINVAL;
}
