// rtg_gmfma.hip — the thin-group k41 layers of the MSD discriminators (retunegan/models/discrminator.py:39-43:
// Conv1d(32, 64, 41, 2, groups=4), (64, 128, 41, 2, groups=8), (128, 512, 41, 4, groups=32), (512, 512, 41, 4, groups=64))
// FORWARD on the matrix cores with tiles that fit the groups exactly (round 4).
//
// The general kernel pads a group's 4-8 input channels to a 16-channel chunk (tap-major: 8-16 output rows of a 32-row
// tile) and reaches 23-35 TFLOP/s on these layers; rtg_gconv.hip moved them to the vector ALUs (packed fp32 FMA, 60-69
// TFLOP/s: scalar-load latency per block of weights, 2-3 waves per SIMD).  Here one v_mfma_f32_16x16x4_f32 tile IS a group:
//   rows     the group's 16 output channels (8 for the last layer: half a tile),
//   columns  16 consecutive output positions,
//   k        4 input channels (lane group kgrp <-> channel 4 c' + kgrp) at ONE tap; four consecutive taps are four
//            matrix instructions, so a lane's B operands of those four are x[ci][p + t .. t + 3] — one aligned 16-byte LDS
//            read (8-byte pairs at stride 2) from the staged input rows, no im2col, no padding channels — and its A operands
//            W[oc][ci][t .. t + 3], 16 bytes of the weight image [group][oc][ci][44] (41 taps padded to 44 with zeros:
//            RTG_PACK_GMFMA_FWD): 93 % of the multiply-adds are real.
// ALL A fragments of a group (Cg / 4 x 11 fragments = 88 registers at 8 channels per group) stay in registers while the
// block walks its work items (clip, block of 64 * NT positions) of that group; per item the Cg input rows of the window go
// through LDS once (coalesced loads, leaky-relu applied once per sample); a wave multiplies NT column tiles against each
// fragment (NT independent accumulators: no dependent matrix instructions back to back).  Per 16 matrix instructions: 4
// LDS reads, nothing else.
// Accumulation order per output: channel quads outermost, taps inside, one fused multiply-add chain — NOT the order of the
// general kernel or of rtg_gconv.hip (rounding-level differences; tests/test_gconv_gpu.py holds all three to the oracle).
#include <type_traits>

#include "rtg_common.h"

namespace {

using rsrc_t = __amdgpu_buffer_rsrc_t;
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int GK = 41, GKP = 44, GPAD = 20;

struct GmArgs {
  const float *x, *w, *bias;
  float* out;
  int B, groups, L_in, L_out;
  float slope;                 // leaky-relu slope of the input (1 = none)
  int n_qb, n_items, per_block;
};

// MG / CG: output / input channels per group; S: stride; NT: 16-position column tiles per wave (4 waves per block)
template <int MG, int CG, int S, int NT>
__global__ __launch_bounds__(256, (CG == 8 && NT == 4) ? 2 : 3) void gmfma_fwd_kernel(const GmArgs a) {
  constexpr int QB = 4 * NT * 16;                    // output positions per work item
  constexpr int WIN = (QB - 1) * S + GKP;            // input samples a work item reads per channel (taps 41 .. 43: zero weights)
  // row pitch (the four channel rows a wave reads at once — lane groups kgrp = 0 .. 3 — are WINP floats apart).  A
  // ds_read_b128 is served 16 lanes at a time, columns 0-3 / 12-15 of one kgrp with columns 4-11 of the next
  // (MI355X_MICROARCH.md): at stride 4 a kgrp's 16 reads are 256 contiguous bytes, so rows a multiple of 256 bytes apart
  // make every such group hit all 64 banks once.  At stride 2 the reads are 8-byte pairs, 32 lanes (two kgrp x 128
  // contiguous bytes) at a time: rows 128 bytes apart modulo 256.
  constexpr int WINP = ((WIN + 63) & ~63) + (S % 4 == 0 ? 0 : 32);
  constexpr int NCQ = CG / 4, NTG = GKP / 4;         // channel quads, tap groups of four
  constexpr int NLD = (CG * WIN + 255) / 256;        // samples a thread stages per work item
  __shared__ __attribute__((aligned(16))) float xs2[2][CG * WINP];     // double buffered over the work items
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n16 = lane & 15, kgrp = lane >> 4;
  const int c_in = a.groups * CG, c_out = a.groups * MG;
  const rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.B * c_in * a.L_in * 4, 0x00020000);
  const int per_g = a.B * a.n_qb;
  const int it0 = blockIdx.x * a.per_block;
  const int it1 = it0 + a.per_block < a.n_items ? it0 + a.per_block : a.n_items;
  // staging: element e = tid + 256 i of the CG x WIN window (WIN is a compile-time constant: the division is a multiply)
  float st[NLD];
  // request the window of `item` (zero outside the row: out-of-range offsets)
  auto stage_issue = [&](int item) __attribute__((always_inline)) {
    const int g = item / per_g;
    const int rest = item - g * per_g;
    const int clip = rest / a.n_qb, qb = rest - clip * a.n_qb;
    const int e0 = qb * QB * S - GPAD;
    const unsigned rowb = (unsigned)((clip * c_in + g * CG) * a.L_in) * 4u;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int e = tid + 256 * i;
      const int ci = e / WIN, w = e - ci * WIN;
      const int pos = e0 + w;
      const unsigned off = (e < CG * WIN && pos >= 0 && pos < a.L_in) ? rowb + (unsigned)(ci * a.L_in + pos) * 4u : 0x80000000u;
      st[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, off, 0, 0));
    }
  };
  auto stage_write = [&](float* xs) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int e = tid + 256 * i;
      const int ci = e / WIN, w = e - ci * WIN;
      float v = st[i];
      v = v > 0.f ? v : v * a.slope;
      if (e < CG * WIN) xs[ci * WINP + w] = v;
    }
  };
  f32x4 af[NCQ][NTG];
  int g_cur = -1, cur = 0;
  if (it0 < it1) {
    stage_issue(it0);
    stage_write(xs2[0]);
  }
  __syncthreads();
  for (int item = it0; item < it1; ++item) {
    // items ordered (group, clip, position block): a block's items share a group but for one change at most
    const int g = item / per_g;
    const int rest = item - g * per_g;
    const int clip = rest / a.n_qb, qb = rest - clip * a.n_qb;
    const int q0 = qb * QB;
    if (g != g_cur) {
      g_cur = g;
      // this lane's A fragments: row m = n16 of the group (rows >= MG: zeros), channel 4 c' + kgrp, taps 4 tg .. + 3
#pragma unroll
      for (int cq = 0; cq < NCQ; ++cq)
#pragma unroll
        for (int tg = 0; tg < NTG; ++tg) {
          f32x4 v = {0.f, 0.f, 0.f, 0.f};
          if (n16 < MG)
            v = *reinterpret_cast<const f32x4*>(a.w + ((size_t)((g * MG + n16) * CG + 4 * cq + kgrp)) * GKP + 4 * tg);
          af[cq][tg] = v;
        }
    }
    // the next item's window: requested now, in flight during this item's matrix instructions, written to the other buffer
    // after them
    if (item + 1 < it1) stage_issue(item + 1);
    const float* xs = xs2[cur];
    // ---- NT column tiles per wave against every fragment
    f32x4 acc[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* xb = xs + kgrp * WINP + ((wave * NT) * 16 + n16) * S;
#pragma unroll
    for (int cq = 0; cq < NCQ; ++cq) {
#pragma unroll
      for (int tg = 0; tg < NTG; ++tg) {
        f32x4 b[NT];
#pragma unroll
        for (int i = 0; i < NT; ++i) {
          const float* p = xb + (4 * cq) * WINP + i * 16 * S + 4 * tg;
          if constexpr (S % 4 == 0) {
            b[i] = *reinterpret_cast<const f32x4*>(p);
          } else {
            const f32x2 lo = *reinterpret_cast<const f32x2*>(p), hi = *reinterpret_cast<const f32x2*>(p + 2);
            b[i] = f32x4{lo.x, lo.y, hi.x, hi.y};
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int i = 0; i < NT; ++i)
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[cq][tg][j], b[i][j], acc[i], 0, 0, 0);
      }
    }
    // ---- store: lane (kgrp, n16) holds rows 4 kgrp .. + 3 of column n16
    float bv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[r] = (a.bias && 4 * kgrp + r < MG) ? a.bias[g * MG + 4 * kgrp + r] : 0.f;
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const int q = q0 + (wave * NT + i) * 16 + n16;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = 4 * kgrp + r;
        if (m < MG && q < a.L_out) a.out[((size_t)clip * c_out + g * MG + m) * a.L_out + q] = acc[i][r] + bv[r];
      }
    }
    if (item + 1 < it1) stage_write(xs2[cur ^ 1]);   // (its last readers passed the barrier of the previous item)
    __syncthreads();
    cur ^= 1;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The last grouped layer, Conv1d(512, 512, 41, 4, groups=64) (discrminator.py:43): 8 input and 8 OUTPUT channels per group —
// half of every 16-row tile would be padding.  Two neighbouring output positions fill the tile instead: row (r, oc) of the
// tile is output channel oc at the positions of parity r, column q' the position pair (2 q', 2 q' + 1):
//   out[oc][2 q' + r] = sum over (ci, u) of W'[(r, oc)][ci][u] * x[ci][8 q' + u - 20],   W'[(r, oc)][ci][u] = w[oc][ci][u - 4 r]
// — a stride-8 layer of 16 rows with 45 taps (padded to 48: the image [group][16][ci][48] of RTG_PACK_GMFMA_FWD with
// RtgPackJob.KH = 4), 41 of 48 multiply-adds real.  Rows of this layer are short (128 / 64 / 32 positions at 8192-sample
// clips = 64 / 32 / 16 pairs): a work item is four 16-pair column tiles, one per wave, TPC of them per clip and 4 / TPC clips
// side by side, so that no wave multiplies an empty tile.  (rtg_gconv.hip ran this layer at 14-53 TFLOP/s: ~50 us per
// launch whatever the row length — a wave per SIMD waiting for its scalar weight loads.)
// VEC: rows whose length is a multiple of 4 are staged with 16-byte loads and LDS writes (a window starts 20 samples left of
// a multiple of 8 and its segments are multiples of 8 long: every quad lies inside the row or outside) — a quarter of the
// staging instructions, which otherwise cost a wave as many issue cycles as its matrix instructions
template <int TPC, bool VEC>
__global__ __launch_bounds__(256, 2) void gmfma_pair_kernel(const GmArgs a) {
  constexpr int CG = 8, MG = 8, SV = 8, KPV = 48, CPI = 4 / TPC;
  constexpr int NCQ = CG / 4, NTG = KPV / 4;
  constexpr int WSEG = (TPC * 16 - 1) * SV + KPV;      // input samples of one clip's TPC tiles per channel
  constexpr int WTOT = CPI * WSEG;
  constexpr int WINP = (WTOT + 63) & ~63;              // (rows a multiple of 256 bytes apart: see gmfma_fwd_kernel)
  constexpr int EW = VEC ? 4 : 1;                      // samples per staged element
  constexpr int NLD = (CG * WTOT / EW + 255) / 256;
  static_assert(WSEG % 4 == 0 && GPAD % 4 == 0, "16-byte staging");
  __shared__ __attribute__((aligned(16))) float xs2[2][CG * WINP];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n16 = lane & 15, kgrp = lane >> 4;
  const int c_in = a.groups * CG, c_out = a.groups * MG;
  const rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.B * c_in * a.L_in * 4, 0x00020000);
  const int csets = (a.B + CPI - 1) / CPI;
  const int per_g = csets * a.n_qb;
  const int it0 = blockIdx.x * a.per_block;
  const int it1 = it0 + a.per_block < a.n_items ? it0 + a.per_block : a.n_items;
  using stage_t = std::conditional_t<VEC, f32x4, float>;
  stage_t st[NLD];
  // element e of the item's CG x WTOT window (in units of EW samples) -> (channel, sample within the row of segments): the
  // same for every item
  int e_ci[NLD], e_w[NLD];
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int e = (tid + 256 * i) * EW;
    e_ci[i] = e / WTOT;
    e_w[i] = e - e_ci[i] * WTOT;
  }
  auto stage_issue = [&](int item) __attribute__((always_inline)) {
    const int g = item / per_g;
    const int rest = item - g * per_g;
    const int cset = rest / a.n_qb, qb = rest - cset * a.n_qb;
    const int e0 = qb * (TPC * 16) * SV - GPAD;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int seg = e_w[i] / WSEG, ws = e_w[i] - seg * WSEG;
      const int clip = cset * CPI + seg, pos = e0 + ws;
      const bool ok = e_ci[i] < CG && clip < a.B && pos >= 0 && pos < a.L_in;
      const unsigned off = ok ? (unsigned)((clip * c_in + g * CG + e_ci[i]) * a.L_in + pos) * 4u : 0x80000000u;
      if constexpr (VEC) st[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0));
      else st[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, off, 0, 0));
    }
  };
  auto stage_write = [&](float* xs) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      if (e_ci[i] >= CG) continue;
      if constexpr (VEC) {
        f32x4 v = st[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = v[k] > 0.f ? v[k] : v[k] * a.slope;
        *reinterpret_cast<f32x4*>(xs + e_ci[i] * WINP + e_w[i]) = v;
      } else {
        float v = st[i];
        xs[e_ci[i] * WINP + e_w[i]] = v > 0.f ? v : v * a.slope;
      }
    }
  };
  f32x4 af[NCQ][NTG];
  int g_cur = -1, cur = 0;
  if (it0 < it1) {
    stage_issue(it0);
    stage_write(xs2[0]);
  }
  __syncthreads();
  const int seg = wave / TPC, tl = wave - seg * TPC;   // this wave's clip of the set and its column tile within the clip
  for (int item = it0; item < it1; ++item) {
    const int g = item / per_g;
    const int rest = item - g * per_g;
    const int cset = rest / a.n_qb, qb = rest - cset * a.n_qb;
    if (g != g_cur) {
      g_cur = g;
      // A fragments: row n16 = (parity, oc) of the group's pair image, channel 4 c' + kgrp, shifted taps 4 tg .. + 3
#pragma unroll
      for (int cq = 0; cq < NCQ; ++cq)
#pragma unroll
        for (int tg = 0; tg < NTG; ++tg)
          af[cq][tg] = *reinterpret_cast<const f32x4*>(a.w + ((size_t)((g * 16 + n16) * CG + 4 * cq + kgrp)) * KPV + 4 * tg);
    }
    if (item + 1 < it1) stage_issue(item + 1);
    const float* xb = xs2[cur] + kgrp * WINP + seg * WSEG + (tl * 16 + n16) * SV;
    // one accumulator per channel quad: two independent chains of 48 matrix instructions
    f32x4 acc[NCQ];
#pragma unroll
    for (int cq = 0; cq < NCQ; ++cq) acc[cq] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tg = 0; tg < NTG; ++tg) {
      f32x4 b[NCQ];
#pragma unroll
      for (int cq = 0; cq < NCQ; ++cq) b[cq] = *reinterpret_cast<const f32x4*>(xb + (4 * cq) * WINP + 4 * tg);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int cq = 0; cq < NCQ; ++cq)
          acc[cq] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[cq][tg][j], b[cq][j], acc[cq], 0, 0, 0);
    }
    // ---- store: lane (kgrp, n16) holds rows 4 kgrp .. + 3 = (parity kgrp / 2, oc 4 (kgrp % 2) + r) of pair n16
    const int clip = cset * CPI + seg;
    const int q = 2 * ((qb * TPC + tl) * 16 + n16) + (kgrp >> 1);
    if (clip < a.B && q < a.L_out) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int oc = 4 * (kgrp & 1) + r;
        float v = acc[0][r];
#pragma unroll
        for (int cq = 1; cq < NCQ; ++cq) v += acc[cq][r];
        a.out[((size_t)clip * c_out + g * MG + oc) * a.L_out + q] = v + (a.bias ? a.bias[g * MG + oc] : 0.f);
      }
    }
    if (item + 1 < it1) stage_write(xs2[cur ^ 1]);
    __syncthreads();
    cur ^= 1;
  }
}

template <int TPC>
int launch_pair(const RtgGconvDesc* d, GmArgs a, hipStream_t s) {
  const bool vec = d->L_in % 4 == 0 && (reinterpret_cast<uintptr_t>(a.x) & 15) == 0;
  const int lv = (d->L_out + 1) / 2;                   // position pairs per row
  a.n_qb = rtg_ceil_div(lv, TPC * 16);
  const long long items = (long long)d->groups * rtg_ceil_div(d->B, 4 / TPC) * a.n_qb;
  if (items > (1ll << 30)) return RTG_ERANGE;
  a.n_items = (int)items;
  long long blocks = items < 1024 ? items : 1024;
  a.per_block = (int)((items + blocks - 1) / blocks);
  blocks = (items + a.per_block - 1) / a.per_block;
  if (vec) RTG_KLAUNCH((gmfma_pair_kernel<TPC, true>), dim3((unsigned)blocks), dim3(256), 0, s, a);
  else RTG_KLAUNCH((gmfma_pair_kernel<TPC, false>), dim3((unsigned)blocks), dim3(256), 0, s, a);
  return rtg_launch_status();
}

// instance serving the problem: 1 = (16, 8, s2), 2 = (16, 4, s4), 3 = (8, 8, s4: position pairs); 0 = none
int gmfma_kind(const RtgGconvDesc* d) {
  if (!d || d->K != GK || d->B < 1 || d->groups < 1 || d->L_in < 1 || d->L_out < 16 || d->pad != GPAD) return 0;
  if ((long long)d->B * d->groups * (d->Cg > d->Mg ? d->Cg : d->Mg) * (d->L_in > d->L_out ? d->L_in : d->L_out) * 4 >=
      (1ll << 31))
    return 0;
  if (d->L_out != (d->L_in + 2 * d->pad - (GK - 1) - 1) / d->stride + 1) return 0;
  if (d->Mg == 16 && d->Cg == 8 && d->stride == 2) return 1;
  if (d->Mg == 16 && d->Cg == 4 && d->stride == 4) return 2;
  if (d->Mg == 8 && d->Cg == 8 && d->stride == 4) return 3;       // (the last grouped layer: gmfma_pair_kernel)
  return 0;
}

template <int MG, int CG, int S, int NT>
int launch_nt(const RtgGconvDesc* d, GmArgs a, hipStream_t s) {
  constexpr int QB = 4 * NT * 16;
  a.n_qb = rtg_ceil_div(d->L_out, QB);
  const long long items = (long long)d->groups * d->B * a.n_qb;
  if (items > (1ll << 30)) return RTG_ERANGE;
  a.n_items = (int)items;
  // about three blocks per CU in flight; a block's items are consecutive (one group, rarely two: the fragments stay)
  long long blocks = items < 1024 ? items : 1024;
  a.per_block = (int)((items + blocks - 1) / blocks);
  blocks = (items + a.per_block - 1) / a.per_block;
  RTG_KLAUNCH((gmfma_fwd_kernel<MG, CG, S, NT>), dim3((unsigned)blocks), dim3(256), 0, s, a);
  return rtg_launch_status();
}

template <int MG, int CG, int S>
int launch_kind(const RtgGconvDesc* d, const GmArgs& a, hipStream_t s) {
  // rows of fewer than 256 positions: one column tile per wave (64 positions per work item)
  if (d->L_out >= 192) return launch_nt<MG, CG, S, 4>(d, a, s);
  return launch_nt<MG, CG, S, 1>(d, a, s);
}

// ---------------------------------------------------------------------------------------------------------------
// backward-weight of the same layers on the matrix cores (round 4; shape code 15 of RtgWgradDesc.shape_cfg):
//   dW[oc][ci][t] = sum over (clip, q) of dy[oc][q] * act(x[ci][q * S + t - 20]),   db[oc] = sum dy[oc][q]
// One v_mfma_f32_16x16x4_f32 tile = the group's 16 output channels x 16 of its Cg * 41 (input channel, tap) columns, k = four
// consecutive positions.  A lane's A operands of four matrix instructions are dy[oc][q .. q + 3]: one 16-byte load straight
// from global memory (every dy element is read by exactly one wave); its B operands x[ci][(q + j) * S + t - 20] are four
// LDS reads S floats apart from the wave's staged input rows — no im2col image: a column's reads walk the row, neighbouring
// columns (taps) neighbouring addresses.  Column Cg * 41 is the bias (a row of ones), the padding columns read a row of
// zeros.  All Cg * 41 / 16 (+1) accumulator tiles of the group (21 x 4 registers at 8 channels per group) stay in registers
// while the wave walks its share of the (clip, position) sequence in blocks of 64 positions; every wave is on its own (own
// LDS rows, no block barrier) and writes one split partial in the bank's layout.  The vector-ALU kernel of rtg_gconv.hip
// (shape code 9) reaches 61-64 TFLOP/s on the large layers (its dy operand is an SGPR pair per two positions: scalar-load
// latency per sub-block); the tuner times both.
// ---------------------------------------------------------------------------------------------------------------
struct GwmArgs {
  const float *x, *dy;
  float* part;
  int B, groups, L_in, L_out;
  int W, bpc, n_blocks, per;                // waves per group, position blocks per clip, blocks in all, blocks per wave
  int red;                                  // 1: the four waves of a block add their tiles in LDS and write ONE partial
  long long part_stride;
  float slope, gy_scale;
};

template <int CG, int S>
__global__ __launch_bounds__(256, 2) void gmfma_wgrad_kernel(const GwmArgs a) {
  constexpr int MG = 16;
  constexpr int N = CG * GK;                         // weight columns of a group; column N: the bias
  constexpr int NCT = (N + 1 + 15) / 16;             // column tiles
  constexpr int PB = 64, NSTEP = PB / 16;            // positions per staged block, 16-position steps
  constexpr int ROWW = (((PB - 1) * S + GK) + 3) & ~3;
  constexpr int ONES = CG * ROWW, ZEROS = ONES + ROWW;
  constexpr int NLD = (CG * ROWW + 63) / 64;
  __shared__ __attribute__((aligned(16))) float xs[4][ZEROS + ROWW];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n16 = lane & 15, kgrp = lane >> 4;
  const int wi = blockIdx.x * 4 + wave;
  if (wi >= a.groups * a.W) return;                  // (no block barriers below: every wave is on its own)
  const int split = wi % a.W, g = wi / a.W;
  const int c_in = a.groups * CG, c_out = a.groups * MG;
  float* xw = xs[wave];
  for (int k = lane; k < ROWW; k += 64) { xw[ONES + k] = 1.f; xw[ZEROS + k] = 0.f; }
  // this lane's column of every tile: LDS offset of (channel, tap) at position 0 of the block, + the lane's k offset
  int colbase[NCT];
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct) {
    const int c = ct * 16 + n16;
    const int ci = c / GK, t = c - ci * GK;
    colbase[ct] = (c < N ? ci * ROWW + t : (c == N ? ONES : ZEROS)) + 4 * kgrp * S;
  }
  f32x4 acc[NCT];
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
  const rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.B * c_in * a.L_in * 4, 0x00020000);
  const rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, a.B * c_out * a.L_out * 4, 0x00020000);
  const int blk0 = split * a.per;
  int blk1 = blk0 + a.per;
  if (blk1 > a.n_blocks) blk1 = a.n_blocks;
  float st[NLD];
  f32x4 an[NSTEP];                                   // the next block's dy fragments (this lane: row n16, positions 4 kgrp ..)
  auto fetch = [&](int blk) __attribute__((always_inline)) {
    const int b = blk / a.bpc, q0 = (blk - b * a.bpc) * PB;
    const int e0 = q0 * S - GPAD;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int e = lane + 64 * i;
      const int ci = e / ROWW, k = e - ci * ROWW;
      const int idx = e0 + k;
      const bool ok = e < CG * ROWW && idx >= 0 && idx < a.L_in;
      st[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
          rx, ok ? (unsigned)((b * c_in + g * CG + ci) * a.L_in + idx) * 4u : 0x80000000u, 0, 0));
    }
    const unsigned rowo = (unsigned)((b * c_out + g * MG + n16) * a.L_out) * 4u;
#pragma unroll
    for (int sp = 0; sp < NSTEP; ++sp) {
      const int q = q0 + sp * 16 + 4 * kgrp;
      an[sp] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rd, q < a.L_out ? rowo + (unsigned)q * 4u : 0x80000000u, 0, 0));
    }
  };
  if (blk0 < blk1) fetch(blk0);
#pragma unroll 1
  for (int blk = blk0; blk < blk1; ++blk) {
    const int b = blk / a.bpc, q0 = (blk - b * a.bpc) * PB;
    (void)b;
    // publish the staged rows (activation applied) and take the dy fragments; elements past the end of the row belong to
    // the next row of dy: zeroed (their x are whatever the window holds)
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int e = lane + 64 * i;
      float v = st[i];
      v = v > 0.f ? v : v * a.slope;
      if (e < CG * ROWW) xw[e] = v;
    }
    f32x4 ac[NSTEP];
#pragma unroll
    for (int sp = 0; sp < NSTEP; ++sp) {
      const int q = q0 + sp * 16 + 4 * kgrp;
#pragma unroll
      for (int j = 0; j < 4; ++j) ac[sp][j] = q + j < a.L_out ? an[sp][j] : 0.f;
    }
    if (blk + 1 < blk1) fetch(blk + 1);               // in flight during this block's matrix instructions
#pragma unroll
    for (int sp = 0; sp < NSTEP; ++sp) {
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) {
        const float* xp = xw + colbase[ct] + sp * 16 * S;
        float bx[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bx[j] = xp[j * S];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[sp][j], bx[j], acc[ct], 0, 0, 0);
      }
    }
  }
  // ---- the four waves of a block are four slices of ONE group's reduction (W a multiple of 4): they meet in LDS, TCH tiles a
  // round, and wave 0 adds them in fixed order (its own, then wave 1, 2, 3) — a quarter of the split partials leave the
  // chip (written here, read back by rtg_weightnorm_backward: at 2048 waves the partials of the four grouped layers of one
  // MSD scale were 130 MB)
  if (a.red) {
    constexpr int TCH = 7;
    static_assert(3 * TCH * 256 <= 4 * (ZEROS + ROWW), "the reduction rounds fit in the staging rows");
    float* red = &xs[0][0];
#pragma unroll
    for (int c0 = 0; c0 < NCT; c0 += TCH) {
      __syncthreads();                                // (the rows' last readers / the previous round's are done)
      if (wave > 0) {
#pragma unroll
        for (int ct = c0; ct < c0 + TCH && ct < NCT; ++ct)
#pragma unroll
          for (int r = 0; r < 4; ++r) red[((wave - 1) * TCH + (ct - c0)) * 256 + r * 64 + lane] = acc[ct][r];
      }
      __syncthreads();
      if (wave == 0) {
#pragma unroll
        for (int ct = c0; ct < c0 + TCH && ct < NCT; ++ct)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float v = acc[ct][r];
#pragma unroll
            for (int w = 0; w < 3; ++w) v += red[(w * TCH + (ct - c0)) * 256 + r * 64 + lane];
            acc[ct][r] = v;
          }
      }
    }
    if (wave != 0) return;
  }
  // ---- this wave's (block's) split partial: rows g * 16 .. + 15 of [rows][N] (+ the bias vector behind the matrix)
  float* wpart = a.part + (size_t)(a.red ? split >> 2 : split) * a.part_stride;
  float* bpart = wpart + (size_t)c_out * N;
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct) {
    const int c = ct * 16 + n16;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = g * MG + 4 * kgrp + r;
      const float v = acc[ct][r] * a.gy_scale;
      if (c < N) wpart[(size_t)row * N + c] = v;
      else if (c == N) bpart[row] = v;
    }
  }
}

int gmfma_wgrad_kind(const RtgWgradDesc* d) {
  if (!d || d->K != GK || d->pad != GPAD || d->dil != 1 || d->C2 != 0 || d->groups < 2 || d->bf16) return 0;
  if (d->h_k > 1 || d->h_n > 1 || d->gy_mode != RTG_PRE_NONE) return 0;
  if (d->pre_mode != RTG_PRE_NONE && d->pre_mode != RTG_PRE_LRELU) return 0;
  if (d->C1 != d->groups * d->Cg || d->dy_L != d->Q || d->Q < 32) return 0;
  if (d->Q != (d->L_in + 2 * GPAD - (GK - 1) - 1) / d->stride + 1) return 0;
  if ((long long)d->B * d->groups * 16 * (d->L_in > d->Q ? d->L_in : d->Q) * 4 >= (1ll << 31)) return 0;
  if (d->Mg == 16 && d->Cg == 8 && d->stride == 2) return 1;
  if (d->Mg == 16 && d->Cg == 4 && d->stride == 4) return 2;
  return 0;
}

}  // namespace

int rtg_gmfma_wgrad_ok(const RtgWgradDesc* d) {
  return gmfma_wgrad_kind(d) > 0 ? 1 : 0;
}

// waves per group: about 2048 waves in all (two per SIMD), at least 4 blocks of 64 positions each; a multiple of 4 where
// there are four or more (the waves of a block then share a group and reduce in LDS)
static int gmfma_wgrad_waves(const RtgWgradDesc* d) {
  const int n_blocks = d->B * rtg_ceil_div(d->Q, 64);
  int w = rtg_ceil_div(2048, d->groups);
  const int w_max = n_blocks / 4 > 0 ? n_blocks / 4 : 1;
  if (w > w_max) w = w_max;
  if (w > 512) w = 512;
  if (w < 1) w = 1;
  if (w >= 4) w &= ~3;
  return w;
}

// split partials the launch writes (every one a whole partial of the layer, read back by rtg_weightnorm_backward): one per
// block of four waves, or one per wave where a group has fewer than four
int rtg_gmfma_wgrad_splits(const RtgWgradDesc* d) {
  if (!gmfma_wgrad_kind(d)) return RTG_EINVAL;
  const int w = gmfma_wgrad_waves(d);
  return w % 4 == 0 ? w / 4 : w;
}

int rtg_gmfma_wgrad_launch(const RtgWgradDesc* d, const float* x, const float* dy, float* part, hipStream_t s) {
  const int kind = gmfma_wgrad_kind(d);
  if (!kind) return RTG_EINVAL;
  if (d->splits != rtg_gmfma_wgrad_splits(d)) return RTG_EINVAL;
  GwmArgs a;
  a.x = x; a.dy = dy; a.part = part;
  a.B = d->B; a.groups = d->groups; a.L_in = d->L_in; a.L_out = d->Q;
  a.W = gmfma_wgrad_waves(d);
  a.red = a.W != d->splits ? 1 : 0;                  // (splits = W / 4: checked above)
  a.bpc = rtg_ceil_div(d->Q, 64);
  a.n_blocks = d->B * a.bpc;
  a.per = rtg_ceil_div(a.n_blocks, a.W);
  a.part_stride = d->part_stride;
  a.slope = d->pre_mode == RTG_PRE_LRELU ? d->pre_slope : 1.f;
  a.gy_scale = d->gy_scale;
  const int waves = d->groups * a.W;
  if (kind == 1) RTG_KLAUNCH((gmfma_wgrad_kernel<8, 2>), dim3(rtg_ceil_div(waves, 4)), dim3(256), 0, s, a);
  else RTG_KLAUNCH((gmfma_wgrad_kernel<4, 4>), dim3(rtg_ceil_div(waves, 4)), dim3(256), 0, s, a);
  return rtg_launch_status();
}

extern "C" int rtg_gmfma_ok(const RtgGconvDesc* d) { return gmfma_kind(d) > 0 ? 1 : 0; }

// floats of the weight image [group][oc][ci][44] (RTG_PACK_GMFMA_FWD); the pair image [group][16][ci][48] (KH = stride)
extern "C" long long rtg_gmfma_workspace(const RtgGconvDesc* d) {
  const int kind = gmfma_kind(d);
  if (kind == 3) return (long long)d->groups * 16 * d->Cg * 48;
  return kind ? (long long)d->groups * d->Mg * d->Cg * GKP : 0;
}

extern "C" int rtg_gmfma_forward(const RtgGconvDesc* d, const float* x, const float* w, const float* bias, float* out,
                                 void* stream) {
  if (!d || !x || !w || !out) return RTG_ENULL;
  const int kind = gmfma_kind(d);
  if (!kind) return RTG_EINVAL;
  if ((reinterpret_cast<uintptr_t>(w) & 15) != 0) return RTG_EINVAL;
  GmArgs a;
  a.x = x; a.w = w; a.bias = bias; a.out = out;
  a.B = d->B; a.groups = d->groups; a.L_in = d->L_in; a.L_out = d->L_out;
  a.slope = d->pre_slope;
  a.n_qb = 0; a.n_items = 0; a.per_block = 0;
  hipStream_t s = (hipStream_t)stream;
  if (kind == 1) return launch_kind<16, 8, 2>(d, a, s);
  if (kind == 3) {
    const int lv = (d->L_out + 1) / 2;
    return lv <= 16 ? launch_pair<1>(d, a, s) : lv <= 32 ? launch_pair<2>(d, a, s) : launch_pair<4>(d, a, s);
  }
  return launch_kind<16, 4, 4>(d, a, s);
}
