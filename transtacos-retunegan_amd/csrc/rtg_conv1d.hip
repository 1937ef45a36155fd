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
//               several clips are packed side by side into one tile ("segments" of seg_len virtual positions), so the
//               MFMA columns are not spent on padding.
//   A operand   weights are pre-packed (rtg_weights_pack) so that one MFMA fragment is 64 consecutive floats:
//               each wave loads its fragments straight from L2 with one coalesced 256-B load, prefetched one
//               (chunk, tap) step ahead — no LDS traffic and no barrier for weights.
//   epilogue    bias, leaky-relu-derivative mask, residual, scale, activation, optional polyphase "shuffle" store.
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "rtg_common.h"

namespace {

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
  // 2-D mode (h_k > 1 or h_n > 1): a "clip" is one (batch item, output row) pair and a "channel" one (channel, kernel
  // row) pair; the kernel row picks which input row of the [items, C, h_in, L_in] tensor the patch row comes from
  int two_d, h_in, h_k, h_stride, h_pad, h_n, h_mode;
  int x_bytes, aux_bytes;
  // tap-major K order (few input channels per group): the k-steps walk (channel, group of KK taps) instead of
  // (tap, group of KK channels); `K` then counts groups of CPN k-steps, K_real the taps, tab_off the LDS offset table
  int tapmajor, K_real, TG, tab_off;
};

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
      const int ci = c / a.h_k, kh = c - ci * a.h_k;
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

template <int TM, int MT, int NT, int MAXIT>   // MAXIT: 64-float pieces of a patch row each lane stages (>= PW / 64)
__global__ __launch_bounds__(RTG_THREADS) void conv1d_mfma_kernel(const ConvArgs a) {
  using M = Mfma<TM>;
  using acc_t = typename M::acc_t;
  constexpr int KK = 64 / TM;           // K-values consumed per MFMA
  constexpr int CPN = RTG_CK / KK;      // MFMA k-steps per (chunk, tap)
  constexpr int RPW = RTG_CK / 4;       // patch rows staged per wave

  extern __shared__ __attribute__((aligned(16))) float lds[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / a.WN, wn = wave - wm * a.WN;
  const int g = blockIdx.y / a.m_blocks, mb = blockIdx.y - g * a.m_blocks;
  const int mt0 = (mb * a.WM + wm) * MT;
  const int BN = a.WN * NT * TM;
  const bool packed = a.seg_len > 0;
  const int b0 = packed ? blockIdx.z * a.seg_nb : blockIdx.z;   // first clip of this block
  const int q_blk = packed ? 0 : blockIdx.x * BN;
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
    if (packed) {
      const int seg = o / a.seg_pitch, w = o - seg * a.seg_pitch;
      bb = b0 + seg;
      pos = w - a.pad;
      ok = ok && seg < a.seg_nb && bb < a.B && w < a.seg_pw;
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
      const int item = (int)eb[it] / a.h_n, hh = (int)eb[it] - item * a.h_n;
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
  const float* wptr[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    int mt = mt0 + i;
    if (mt > a.n_mt - 1) mt = a.n_mt - 1;   // clamped duplicate tile, discarded in the epilogue
    wptr[i] = a.wp + ((size_t)(g * a.n_mt + mt) * a.n_cc) * a.K * (RTG_CK * TM) + lane;
  }
  const int n_steps = a.n_cc * a.K;
  // two named A-fragment register sets used alternately (no loop-carried copy: with a copy at the end of the tap the
  // compiler waits for the JUST-issued prefetch in the middle of the MFMA phase, one exposed L2 latency per tap)
  float a0[MT][CPN], a1[MT][CPN];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int cp = 0; cp < CPN; ++cp) a0[i][cp] = wptr[i][cp * 64];

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
  stage(0);
  swrite(lds);
  __syncthreads();

  int cc = 0, tap = 0;
  // one (chunk, tap) step: prefetch the next step's A fragments into `nxt`, multiply with `cur`
  auto do_step = [&](int step, float (&cur)[MT][CPN], float (&nxt)[MT][CPN]) __attribute__((always_inline)) {
    const float* buf = lds + (cc & 1) * bufsz;
    if (step + 1 < n_steps) {
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int cp = 0; cp < CPN; ++cp) nxt[i][cp] = wptr[i][(size_t)(step + 1) * (RTG_CK * TM) + cp * 64];
    }
    // the next chunk's patch is requested AFTER the weight prefetch, on the chunk's first tap: vmcnt retires in order,
    // so the wait for `nxt` one tap later does not include these loads, the wait two taps later finds them landed
    if (tap == 0 && cc + 1 < a.n_cc) stage(cc + 1);
    const int td = tap * a.dil;
    const int tapoff = (a.stride == 1) ? td : (td % a.stride) * a.PH + td / a.stride;
    const float* bp = buf + bbase + tapoff;
    // read phase: all B fragments of this (chunk, tap) into distinct registers, THEN the MFMA phase — the compiler
    // otherwise recycles one register pair and serialises ds_read -> wait -> 2 MFMAs per k-step; with two waves
    // per SIMD one wave's read phase overlaps the other's MFMA phase
    float bf[CPN][NT];
    if (a.tapmajor) {
      int boff[CPN];
#pragma unroll
      for (int cp = 0; cp < CPN; ++cp) boff[cp] = tab[(tap * CPN + cp) * KK + kk];
      const float* b0p = buf + wn * NT * TM + n_lane;
#pragma unroll
      for (int cp = 0; cp < CPN; ++cp)
#pragma unroll
        for (int j = 0; j < NT; ++j) bf[cp][j] = b0p[boff[cp] + j * TM];
    } else {
#pragma unroll
      for (int cp = 0; cp < CPN; ++cp)
#pragma unroll
        for (int j = 0; j < NT; ++j) bf[cp][j] = bp[cp * KK * a.ROW + j * TM];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int cp = 0; cp < CPN; ++cp)
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = M::run(cur[i][cp], bf[cp][j], acc[i][j]);
    __builtin_amdgcn_sched_barrier(0);
    if (++tap == a.K) {
      tap = 0;
      if (cc + 1 < a.n_cc) swrite(lds + ((cc + 1) & 1) * bufsz);
      __syncthreads();
      ++cc;
    }
  };
  int step = 0;
  for (; step + 1 < n_steps; step += 2) {
    do_step(step, a0, a1);
    do_step(step + 1, a1, a0);
  }
  if (step < n_steps) do_step(step, a0, a1);

  // ---- epilogue, fast path (plain store): 32-bit element offsets, every optional operand (bias, mask, residual,
  // accumulate) read through a buffer descriptor that has ZERO records when the operand is absent (the load then
  // returns 0 without touching memory), invalid rows / columns stored to an out-of-range offset (dropped by the
  // hardware): no per-element branches, all loads of a tile in flight together, stores issue back to back
  if (a.shuf_S == 1 && a.out_split == 0 && !a.two_d) {
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
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      if (mt0 + i >= a.n_mt) continue;
      const int mbase = (mt0 + i) * TM;
      float bv[M::NREG];
#pragma unroll
      for (int r = 0; r < M::NREG; ++r) {
        const int m = mbase + M::row(lane, r);
        bv[r] = buf_load(rb, m < a.Mg ? (unsigned)(g * a.Mg + m) * 4u : RTG_OOB);
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        int q = q_blk + (wn * NT + j) * TM + n_lane;
        int b = b0;
        bool ok = true;
        if (packed) {
          const int seg = q / a.seg_len;
          q -= seg * a.seg_len;
          b = b0 + seg;
          ok = seg < a.seg_nb && b < a.B;
        }
        ok = ok && q < a.Q;
        const unsigned col = ok ? ((unsigned)(b * a.out_C + g * a.Mg + mbase) * (unsigned)a.out_L + (unsigned)q) * 4u
                                : RTG_OOB;
        unsigned off[M::NREG];
        float mv[M::NREG], rv[M::NREG], av[M::NREG];
#pragma unroll
        for (int r = 0; r < M::NREG; ++r) {
          const int row = M::row(lane, r);
          off[r] = (mbase + row < a.Mg) ? (col + (unsigned)row * (unsigned)a.out_L * 4u) | (col & RTG_OOB) : RTG_OOB;
          mv[r] = buf_load(rm, off[r]);
          rv[r] = buf_load(rr, off[r]);
          av[r] = buf_load(ra, off[r]);
        }
#pragma unroll
        for (int r = 0; r < M::NREG; ++r) {
          float v = acc[i][j][r] + bv[r];
          v *= (mv[r] > 0.f ? 1.f : mslope);
          v = (v + rv[r]) * a.out_scale;
          if (a.act == RTG_ACT_LRELU) v = rtg_lrelu(v, a.act_slope);
          else if (a.act == RTG_ACT_TANH) v = tanhf(v);
          v += av[r];
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ro, off[r], 0, 0);
        }
      }
    }
    return;
  }

  // ---- epilogue, general path (polyphase shuffle store, concat-split store, 2-D outputs)
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    if (mt0 + i >= a.n_mt) continue;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      int q = q_blk + (wn * NT + j) * TM + n_lane;
      int b = b0;
      if (packed) {
        const int seg = q / a.seg_len;
        q -= seg * a.seg_len;
        b = b0 + seg;
        if (seg >= a.seg_nb || b >= a.B) continue;
      }
      if (q >= a.Q) continue;
      int hh = 0;
      if (a.two_d) {                     // clip -> (batch item, output row)
        const int item = b / a.h_n;
        hh = b - item * a.h_n;
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
}

struct TileCfg {
  int MT, NT, WM, WN;
  int seg_len, seg_nb;   // > 0: clips packed per block
};

template <int TM, int MT, int NT, int MAXIT>
int launch_it(const ConvArgs& a, dim3 grid, size_t lds_bytes, hipStream_t s) {
  auto k = conv1d_mfma_kernel<TM, MT, NT, MAXIT>;
  if (lds_bytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return -(1000 + (int)e);
  }
  hipLaunchKernelGGL(k, grid, dim3(RTG_THREADS), lds_bytes, s, a);
  return rtg_launch_status();
}

template <int TM, int MT, int NT>
int launch(const ConvArgs& a, dim3 grid, size_t lds_bytes, hipStream_t s) {
  if (a.PW <= 3 * 64) return launch_it<TM, MT, NT, 3>(a, grid, lds_bytes, s);
  if (a.PW <= 5 * 64) return launch_it<TM, MT, NT, 5>(a, grid, lds_bytes, s);
  return launch_it<TM, MT, NT, RTG_PW_MAX / 64>(a, grid, lds_bytes, s);
}

// Patch width (floats per channel) for a block covering BN output positions.
inline int patch_width(int BN, int stride, int K, int dil) { return (BN - 1) * stride + (K - 1) * dil + 1; }

// virtual positions one clip occupies when clips are packed side by side: its Q outputs plus the gap that keeps the
// next clip's input patch from overlapping (pitch = seg_len * stride >= (Q-1)*stride + (K-1)*dil + 1)
inline int segment_len(int Q, int stride, int K, int dil) {
  const int extra = (K - 1) * dil + 1 - stride;
  return Q + (extra > 0 ? (extra + stride - 1) / stride : 0);
}

// Scores every block shape valid for the problem (score <= 0: not applicable); returns the number of shapes.
constexpr int kMaxTileCfgs = 16;
struct ScoredCfg {
  TileCfg c;
  double score;
};
inline int tile_code(const TileCfg& c) { return c.MT * 100 + c.NT * 10 + c.WM; }

int score_tiles(int TM, int n_mt, int Q, int B, int groups, int stride, int K, int dil, ScoredCfg* out) {
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
    TileCfg c = {cs[i][0], cs[i][1], cs[i][2], cs[i][3], 0, 0};
    const int BN = c.WN * c.NT * TM;
    if (patch_width(BN, stride, K, dil) > RTG_PW_MAX) continue;
    const int mrows = c.WM * c.MT;
    const int m_blocks = rtg_ceil_div(n_mt, mrows);
    double eff_q, n_blocks_q;
    const int nb = BN / Lseg;
    if (nb >= 2 && B >= 2) {             // pack nb clips per block
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
  }
  return cnt;
}

// forced: a RtgConv1dDesc.tile_cfg code, or 0 for the best score.  MT == 0 in the result: nothing applicable.
TileCfg pick_tiles(int TM, int n_mt, int Q, int B, int groups, int stride, int K, int dil, int forced) {
  ScoredCfg sc[kMaxTileCfgs];
  const int n = score_tiles(TM, n_mt, Q, B, groups, stride, K, dil, sc);
  TileCfg best = {0, 0, 0, 0, 0, 0};
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

extern "C" int rtg_conv1d_variant(const RtgConv1dDesc* d) {
  if (!d) return RTG_ENULL;
  if ((d->tile_m != 32 && d->tile_m != 16) || d->Mg < 1 || d->Q < 1 || d->B < 1 || d->groups < 1 || d->stride < 1 ||
      d->K < 1 || d->dil < 1)
    return RTG_EINVAL;
  if (d->tile_cfg == 0 && rtg_thin_kind(d)) return rtg_thin_kind(d);
  const TileCfg c = pick_tiles(d->tile_m, rtg_ceil_div(d->Mg, d->tile_m), d->Q, d->B, d->groups, d->stride, d->K, d->dil,
                               d->tile_cfg);
  if (c.MT == 0) return d->tile_cfg ? RTG_EINVAL : RTG_ERANGE;
  return d->tile_m * 100 + c.MT * 10 + c.NT;
}

extern "C" int rtg_conv1d_tile_candidates(const RtgConv1dDesc* d, int* cfgs, int max) {
  if (!d || !cfgs) return RTG_ENULL;
  if ((d->tile_m != 32 && d->tile_m != 16) || d->Mg < 1 || d->Q < 1 || d->B < 1 || d->groups < 1 || d->stride < 1 ||
      d->K < 1 || d->dil < 1 || max < 1)
    return RTG_EINVAL;
  if (rtg_thin_kind(d)) {       // served by a bandwidth kernel: nothing to choose (0 = the library's default)
    cfgs[0] = 0;
    return 1;
  }
  ScoredCfg sc[kMaxTileCfgs];
  const int n = score_tiles(d->tile_m, rtg_ceil_div(d->Mg, d->tile_m), d->Q, d->B, d->groups, d->stride, d->K, d->dil, sc);
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
  return cnt;
}

extern "C" long long rtg_packed_size(int groups, int Mg, int Cg, int K, int tile_m) {
  if (groups < 1 || Mg < 1 || Cg < 1 || K < 1 || (tile_m != 32 && tile_m != 16)) return RTG_EINVAL;
  const long long n_mt = (Mg + tile_m - 1) / tile_m, n_cc = (Cg + RTG_CK - 1) / RTG_CK;
  return (long long)groups * n_mt * n_cc * K * RTG_CK * tile_m;
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

extern "C" int rtg_conv1d(const RtgConv1dDesc* d, const float* x1, const float* x2, const float* aux, const float* wp,
                          const float* bias, const float* mask, const float* res, float* out, float* out2,
                          void* stream) {
  if (!d || !x1 || !wp) return RTG_ENULL;
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

  if (d->tile_cfg == 0) {
    const int thin = rtg_thin_kind(d);
    if (thin) return rtg_thin_launch(thin, d, x1, aux, wp, bias, mask, res, out, (hipStream_t)stream);
  }

  ConvArgs a;
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

  const TileCfg c = pick_tiles(TM, a.n_mt, d->Q, d->B, d->groups, d->stride, d->K, d->dil, d->tile_cfg);
  if (c.MT == 0) return d->tile_cfg ? RTG_EINVAL : RTG_ERANGE;   // unknown shape / even the smallest patch exceeds RTG_PW_MAX
  a.WM = c.WM; a.WN = c.WN;
  const int BN = c.WN * c.NT * TM;
  a.PW = patch_width(BN, d->stride, d->K, d->dil);
  a.seg_len = c.seg_len; a.seg_nb = c.seg_nb;
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
  const long long gy = (long long)d->groups * a.m_blocks;
  if (gy > 65535) return RTG_ERANGE;
  const int gz = c.seg_len > 0 ? rtg_ceil_div(d->B, c.seg_nb) : d->B;
  if (gz > 65535) return RTG_ERANGE;
  dim3 grid(c.seg_len > 0 ? 1 : rtg_ceil_div(d->Q, BN), (unsigned)gy, gz);
  a.tab_off = 2 * RTG_CK * a.ROW;
  const size_t lds_bytes = (size_t)(2 * RTG_CK * a.ROW + (a.tapmajor ? a.K * RTG_CK : 0)) * sizeof(float);
  hipStream_t s = (hipStream_t)stream;

#define RTG_CASE(tm, mt, nt) \
  if (TM == tm && c.MT == mt && c.NT == nt) return launch<tm, mt, nt>(a, grid, lds_bytes, s);
  RTG_CASE(32, 1, 1) RTG_CASE(32, 1, 2) RTG_CASE(32, 1, 4) RTG_CASE(32, 2, 1) RTG_CASE(32, 2, 2)
  RTG_CASE(16, 1, 1) RTG_CASE(16, 1, 2) RTG_CASE(16, 1, 4)
#undef RTG_CASE
  return RTG_ERANGE;
}
