// rtg_gconv.hip — the thin-group strided convolutions of the MSD discriminators (retunegan/models/discrminator.py:39-43:
// Conv1d(32, 64, 41, 2, groups=4), (64, 128, 41, 2, groups=8), (128, 512, 41, 4, groups=32), (512, 512, 41, 4, groups=64))
// on the vector ALUs.
//
// With 4-8 input and 8-16 output channels per group the matrix-core path multiplies tiles that are mostly padding
// (8 of 16 rows, a 164- or 328-long reduction in tap-major order) and reaches 23-33 TFLOP/s on the stride-4 layers.
// The packed fp32 FMA (v_pk_fma_f32: two FMAs per lane and cycle) has the same peak as the fp32 matrix instruction
// (157 TFLOP/s) and no tile shape: a lane owns P consecutive output positions of ALL output channels of one group,
// keeps the input window of one input channel ((P-1)*S + K samples) in registers, and walks the taps with compile-time register
// indices: per (input channel, tap) ONE scalar load of the group's MG weights (an SGPR operand of the packed FMAs:
// broadcast LDS reads of the weights by four SIMDs saturate the CU's LDS port) feeds P*MG/2 packed FMAs.  The input row
// segment of a work item goes through LDS once (coalesced loads, activation applied once per sample, each wave its own
// double buffer: no block barriers), the lanes' overlapping windows are 16-byte LDS reads.  The effective weights
// g*v/||v|| are laid out as [group][ci][tap][oc] by rtg_gconv_prepare from the raw weight-norm parameters.
//
// Accumulation order: per output, input channels outermost, taps innermost, all in one fp32 FMA chain — NOT the order
// of the matrix-core kernel (rounding-level differences; parity tolerances are those of every other conv test).
#include <cstdlib>
#include <utility>

#include "rtg_common.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
using rsrc_t = __amdgpu_buffer_rsrc_t;
constexpr int GK = 41;                      // taps of every layer this file serves

struct GcArgs {
  int B, groups, L_in, L_out, pad;
  int tiles, csets, n_items;                // position tiles per clip, clip sets, work items (group, clip set, tile)
  float slope;                              // leaky-relu slope applied to the input (1 = none)
};

__device__ __forceinline__ float gload(rsrc_t r, unsigned off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
}

typedef float f32x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void sload(f32x8& w, const float* p) {
  asm volatile("s_load_dwordx8 %0, %1, 0x0" : "=&s"(w) : "s"(p));
}
// the SGPR set is an in/out operand: its users are ordered after the wait
__device__ __forceinline__ void swait(f32x8& w) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(w)); }

// acc += w * x.lo / x.hi (both halves of the result take the same half of x; w is an SGPR pair)
__device__ __forceinline__ void pkfma_lo(f32x2& acc, f32x2 w, f32x2 x) {
  asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "s"(w), "v"(x));
}
__device__ __forceinline__ void pkfma_hi(f32x2& acc, f32x2 w, f32x2 x) {
  asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "s"(w), "v"(x));
}

// w'[g][ci][t][oc] = v[g*MG + oc][ci][t] * scale[g*MG + oc]: a tap's MG weights contiguous, so that a wave fetches them
// with one scalar load
__global__ __launch_bounds__(RTG_THREADS) void gconv_prep_kernel(const float* __restrict__ v, const float* __restrict__ scale,
                                                                 float* __restrict__ w, int MG, int CK, int total) {
  const int e = blockIdx.x * RTG_THREADS + threadIdx.x;          // destination index
  if (e >= total) return;
  const int oc = e % MG, r = (e / MG) % CK, g = e / (MG * CK);
  const int row = g * MG + oc;
  w[e] = v[(size_t)row * CK + r] * scale[row];
}

// backward-data order: w'[g][oc][t][ci] = v[g*MG + oc][ci][t] * scale[g*MG + oc]
__global__ __launch_bounds__(RTG_THREADS) void gconv_prep_bwd_kernel(const float* __restrict__ v,
                                                                     const float* __restrict__ scale, float* __restrict__ w,
                                                                     int CG, int K, int total) {
  const int e = blockIdx.x * RTG_THREADS + threadIdx.x;          // destination index
  if (e >= total) return;
  const int ci = e % CG, t = (e / CG) % K, row = e / (CG * K);
  w[e] = v[((size_t)row * CG + ci) * K + t] * scale[row];
}

// A block of TB = 32 / MG taps (32 weights = four 8-register scalar loads): wait for its weights, request the next
// block's into the other SGPR set, TB * P * MG / 2 packed FMAs.  Scalar loads return out of order, so the only safe wait
// is lgkmcnt(0), which also covers whatever was requested last: requests are therefore made in blocks, right after the
// wait, and have a whole block of FMAs (256 cycles of this wave alone) to land.  A function template per block and a
// fold over the index sequence instead of an unrolled loop: the window's register indices must be compile-time
// constants and the loop unroller gives up on bodies of this size (1300 asm statements).
template <int BI, int MG, int S, int P, int NW>
__device__ __forceinline__ void tap_block(f32x2 (&acc)[MG / 2][P], const f32x2 (&win)[NW], f32x8 (&wq)[2][4],
                                          const float* wrow) {
  constexpr int TB = 32 / MG, NB = (GK + TB - 1) / TB;
#ifndef RTG_GC_NOW
#pragma unroll
  for (int h = 0; h < 4; ++h) swait(wq[BI % 2][h]);
  if (BI + 1 < NB) {
#pragma unroll
    for (int h = 0; h < 4; ++h) sload(wq[(BI + 1) % 2][h], wrow + (BI + 1) * 32 + 8 * h);
  }
  constexpr int SET = BI % 2;
#else                      // ablation: the first block's weights for all taps (no scalar loads in the loop)
  if (BI == 0) {
#pragma unroll
    for (int h = 0; h < 4; ++h) swait(wq[0][h]);
  }
  constexpr int SET = 0;
#endif
#pragma unroll
  for (int tb = 0; tb < TB; ++tb) {
    constexpr int dummy = 0; (void)dummy;
    const int t = BI * TB + tb;
    if (t < GK) {
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const int e = p * S + t;
#pragma unroll
        for (int o = 0; o < MG / 2; ++o) {
          const int wi = tb * MG + 2 * o;               // index of the pair within the block's 32 weights
          const f32x8 wv = wq[SET][wi / 8];
          f32x2 w2;
          w2.x = wv[wi % 8]; w2.y = wv[wi % 8 + 1];
          if (e & 1) pkfma_hi(acc[o][p], w2, win[e / 2]);
          else pkfma_lo(acc[o][p], w2, win[e / 2]);
        }
      }
    }
  }
}

template <int MG, int S, int P, int NW, int... BI>
__device__ __forceinline__ void all_taps(f32x2 (&acc)[MG / 2][P], const f32x2 (&win)[NW], f32x8 (&wq)[2][4],
                                         const float* wrow, std::integer_sequence<int, BI...>) {
  (tap_block<BI, MG, S, P, NW>(acc, win, wq, wrow), ...);
}

// x, w, bias, out as separate restrict parameters: the weight loads have uniform addresses and must become SCALAR loads,
// which the compiler only emits for memory it knows the kernel does not write.
// w: the layer's effective weights as [group][ci][tap][oc] (gconv_prep_kernel)
template <int MG, int CG, int S, int P, int LPC>
__global__ __launch_bounds__(RTG_THREADS) void gconv_fwd_kernel(const float* __restrict__ gx, const float* __restrict__ gw,
                                                                const float* __restrict__ gbias, float* __restrict__ gout,
                                                                const GcArgs a) {
  // LPC lanes work on one clip (LPC * P output positions per tile); rows shorter than 64 * P positions put 64 / LPC
  // clips side by side in a wave, all of the same group (same weights)
  constexpr int CPW = 64 / LPC;             // clips per work item
  constexpr int WIN = (P - 1) * S + GK;     // input samples a lane's P outputs read per input channel
  constexpr int WINP = (WIN + 3) & ~3;      // ... fetched as 16-byte LDS reads, kept as register pairs
  constexpr int SPAN = LPC * P * S;         // input samples between the first positions of consecutive tiles
  constexpr int ROWF = SPAN + WINP;         // staged samples per clip and input channel (+ tail the last lane reads)
  constexpr int NLD = (CPW * ROWF + 63) / 64;   // samples a lane stages
  __shared__ __attribute__((aligned(16))) float xs[4][2][CPW * ROWF];  // per wave, double buffered over the input channels
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int sub = lane / LPC, ll = lane - sub * LPC;
  const int c_in = a.groups * CG, c_out = a.groups * MG;
  const bool act = a.slope != 1.f;
  // the whole input tensor through one descriptor: the clips of a work item have different row bases, so the offsets are
  // per lane; padding and missing clips are requested at an out-of-range offset (returned as 0)
  const rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)gx, 0, a.B * c_in * a.L_in * 4, 0x00020000);
  for (int item = blockIdx.x * 4 + wave; item < a.n_items; item += gridDim.x * 4) {
    // items ordered (group, clip set, tile): the waves of a block and neighbouring blocks share a group's weights
    const int per_g = a.csets * a.tiles;
    const int g = item / per_g;
    const int rest = item - g * per_g;
    const int cset = rest / a.tiles, tile = rest - cset * a.tiles;
    const int clip = cset * CPW + sub;                // this lane's clip (may be past the batch)
    const int q0 = (tile * LPC + ll) * P;             // the lane's first output position
    f32x2 acc[MG / 2][P];
#pragma unroll
    for (int o = 0; o < MG / 2; ++o) {
      f32x2 b;
      b.x = gbias ? gbias[g * MG + 2 * o] : 0.f;
      b.y = gbias ? gbias[g * MG + 2 * o + 1] : 0.f;
#pragma unroll
      for (int p = 0; p < P; ++p) acc[o][p] = b;
    }
    const int e_tile = tile * SPAN - a.pad;           // input index of the tile's first staged sample
    const float* wg = gw + (size_t)g * (CG * GK * MG);
    // staging geometry (per lane, the same for every input channel): element j of the item's CPW * ROWF samples
    unsigned soff[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int j = lane + 64 * i;
      const int sc = j / ROWF, w = j - sc * ROWF;     // (clip within the item, sample within its row segment)
      const int e = e_tile + w;
      const int c = cset * CPW + sc;
      const bool ok = j < CPW * ROWF && e >= 0 && e < a.L_in && c < a.B;
      soff[i] = ok ? (unsigned)((c * c_in + g * CG) * a.L_in + e) * 4u : 0x80000000u;
    }
    float st[NLD];
    auto fetch = [&](int ci) __attribute__((always_inline)) {
      const unsigned coff = (unsigned)(ci * a.L_in) * 4u;
#pragma unroll
      for (int i = 0; i < NLD; ++i) st[i] = gload(rx, soff[i] + coff);      // (the out-of-range bit survives the add)
    };
    auto publish = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < NLD; ++i) {
        const int j = lane + 64 * i;
        float v = st[i];
        if (act) v = v > 0.f ? v : v * a.slope;
        if (j < CPW * ROWF) xs[wave][buf][j] = v;
      }
    };
    fetch(0);
    publish(0);
#pragma unroll 1
    for (int ci = 0; ci < CG; ++ci) {
      if (ci + 1 < CG) fetch(ci + 1);                 // in flight during this channel's FMAs
      // the lane's window: LDS reads of the wave's own writes (same wave: in order, no barrier)
      f32x2 win[WINP / 2];                            // sample e = half (e & 1) of pair e / 2
      const float* xw = &xs[wave][ci & 1][sub * ROWF + ll * (P * S)];
#pragma unroll
      for (int j = 0; j < WINP; j += 4) {
        const f32x4 q = *reinterpret_cast<const f32x4*>(xw + j);
        win[j / 2].x = q.x; win[j / 2].y = q.y; win[j / 2 + 1].x = q.z; win[j / 2 + 1].y = q.w;
      }
      const float* wrow = wg + ci * (GK * MG);
      // the weights come in blocks of 32 (tap_block); requests, waits AND the FMAs are volatile asm: the instruction
      // selector otherwise issues all requests first and the FMAs after them (640 SGPRs spilled to vector lanes)
      f32x8 wq[2][4];
#pragma unroll
      for (int h = 0; h < 4; ++h) sload(wq[0][h], wrow + 8 * h);
      all_taps<MG, S, P, WINP / 2>(acc, win, wq, wrow, std::make_integer_sequence<int, (GK + 32 / MG - 1) / (32 / MG)>{});
      if (ci + 1 < CG) publish((ci + 1) & 1);
    }
    // ---- store: P consecutive positions per (lane, output channel)
    if (clip < a.B) {
      float* orow = gout + ((size_t)clip * c_out + (size_t)g * MG) * a.L_out;
      const bool vec = P % 4 == 0 && (a.L_out & 3) == 0 && (((size_t)gout) & 15) == 0;
#pragma unroll
      for (int o = 0; o < MG / 2; ++o) {
        if (vec) {                                    // 16-byte stores: L_out a multiple of 4, q0 a multiple of P
#pragma unroll
          for (int p = 0; p + 3 < P; p += 4) {
            if (q0 + p < a.L_out) {
              f32x4 v0, v1;
              v0.x = acc[o][p].x; v0.y = acc[o][p + 1].x; v0.z = acc[o][p + 2].x; v0.w = acc[o][p + 3].x;
              v1.x = acc[o][p].y; v1.y = acc[o][p + 1].y; v1.z = acc[o][p + 2].y; v1.w = acc[o][p + 3].y;
              *reinterpret_cast<f32x4*>(orow + (size_t)(2 * o) * a.L_out + q0 + p) = v0;
              *reinterpret_cast<f32x4*>(orow + (size_t)(2 * o + 1) * a.L_out + q0 + p) = v1;
            }
          }
        } else {
#pragma unroll
          for (int p = 0; p < P; ++p) {
            if (q0 + p < a.L_out) {
              orow[(size_t)(2 * o) * a.L_out + q0 + p] = acc[o][p].x;
              orow[(size_t)(2 * o + 1) * a.L_out + q0 + p] = acc[o][p].y;
            }
          }
        }
      }
    }
  }
}

template <int MG, int CG, int S, int P, int LPC>
int launch_lpc(const RtgGconvDesc* d, const float* x, const float* w, const float* bias, float* out, const GcArgs& a0,
               hipStream_t s) {
  GcArgs a = a0;
  a.tiles = rtg_ceil_div(d->L_out, LPC * P);
  a.csets = rtg_ceil_div(d->B, 64 / LPC);
  const long long items = (long long)d->groups * a.csets * a.tiles;
  if (items > (1ll << 30)) return RTG_ERANGE;
  a.n_items = (int)items;
  int blocks = rtg_ceil_div(items, 4);
  if (blocks > 2048) blocks = 2048;                   // a wave then walks consecutive items of (mostly) one group
  RTG_KLAUNCH((gconv_fwd_kernel<MG, CG, S, P, LPC>), dim3(blocks), dim3(RTG_THREADS), 0, s, x, w, bias, out, a);
  return rtg_launch_status();
}

// rows shorter than a full wave's tile: several clips per wave
template <int MG, int CG, int S, int P>
int launch_fwd(const RtgGconvDesc* d, const float* x, const float* w, const float* bias, float* out, const GcArgs& a,
               hipStream_t s) {
  const int need = rtg_ceil_div(d->L_out, P);         // lanes a whole row takes
  if (need > 32) return launch_lpc<MG, CG, S, P, 64>(d, x, w, bias, out, a, s);
  if (need > 16) return launch_lpc<MG, CG, S, P, 32>(d, x, w, bias, out, a, s);
  if (need > 8) return launch_lpc<MG, CG, S, P, 16>(d, x, w, bias, out, a, s);
  return launch_lpc<MG, CG, S, P, 8>(d, x, w, bias, out, a, s);
}

// ---------------------------------------------------------------------------------------------------------------
// backward-data: dx[ci][i] = res[ci][i] + mask(x[ci][i]) * sum over (oc, tap t with (i + PAD - t) % S == 0) of
// w[oc][ci][t] * dy[oc][(i + PAD - t) / S].  Same structure with the roles swapped: a lane owns P consecutive INPUT
// positions (P a multiple of S, first position a multiple of S: which taps reach which position is then a compile-time
// pattern) of all CG input channels of a group, the reduction runs over the MG output channels with the window of
// dy[oc] in registers, the weights are [group][oc][tap][ci] (rtg_gconv_prepare_bwd) in blocks of 32 = 32 / CG taps.
// ---------------------------------------------------------------------------------------------------------------
constexpr int GPAD = 20;                    // padding of every layer this file serves

template <int BI, int CG, int S, int P, int NW>
__device__ __forceinline__ void tap_block_bwd(f32x2 (&acc)[CG / 2][P], const f32x2 (&win)[NW], f32x8 (&wq)[2][4],
                                              const float* wrow) {
  constexpr int TB = 32 / CG, NB = (GK + TB - 1) / TB;
  constexpr int M0 = -((GK - 1 - GPAD + S - 1) / S);        // floor((GPAD - (GK - 1)) / S): first window element's offset
#pragma unroll
  for (int h = 0; h < 4; ++h) swait(wq[BI % 2][h]);
  if (BI + 1 < NB) {
#pragma unroll
    for (int h = 0; h < 4; ++h) sload(wq[(BI + 1) % 2][h], wrow + (BI + 1) * 32 + 8 * h);
  }
#pragma unroll
  for (int tb = 0; tb < TB; ++tb) {
    const int t = BI * TB + tb;
    if (t < GK) {
#pragma unroll
      for (int p = 0; p < P; ++p) {
        if ((p + GPAD - t + 64 * S) % S == 0) {            // tap t reaches position p of the lane
          const int e = (p + GPAD - t + 64 * S) / S - 64 - M0;   // window element: dy position (p + GPAD - t) / S
#pragma unroll
          for (int c = 0; c < CG / 2; ++c) {
            const int wi = tb * CG + 2 * c;
            const f32x8 wv = wq[BI % 2][wi / 8];
            f32x2 w2;
            w2.x = wv[wi % 8]; w2.y = wv[wi % 8 + 1];
            if (e & 1) pkfma_hi(acc[c][p], w2, win[e / 2]);
            else pkfma_lo(acc[c][p], w2, win[e / 2]);
          }
        }
      }
    }
  }
}

template <int CG, int S, int P, int NW, int... BI>
__device__ __forceinline__ void all_taps_bwd(f32x2 (&acc)[CG / 2][P], const f32x2 (&win)[NW], f32x8 (&wq)[2][4],
                                             const float* wrow, std::integer_sequence<int, BI...>) {
  (tap_block_bwd<BI, CG, S, P, NW>(acc, win, wq, wrow), ...);
}

template <int MG, int CG, int S, int P, int LPC>
__global__ __launch_bounds__(RTG_THREADS) void gconv_bwd_kernel(const float* __restrict__ gdy, const float* __restrict__ gw,
                                                                const float* __restrict__ gmask, const float* __restrict__ gres,
                                                                float* __restrict__ gdx, const GcArgs a) {
  static_assert(P % S == 0 && P % 4 == 0, "a lane's positions cover whole phase periods");
  constexpr int CPW = 64 / LPC;
  constexpr int M0 = -((GK - 1 - GPAD + S - 1) / S);
  constexpr int WIN = (P - 1 + GPAD) / S - M0 + 1;          // dy samples a lane's P positions read per output channel
  constexpr int WINP = (WIN + 3) & ~3;
  constexpr int SPAN = LPC * P / S;                         // dy samples between consecutive tiles
  constexpr int ROWF = ((SPAN + WINP) + 3) & ~3;
  constexpr int NLD = (CPW * ROWF + 63) / 64;
  __shared__ __attribute__((aligned(16))) float ys[4][2][CPW * ROWF];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int sub = lane / LPC, ll = lane - sub * LPC;
  const int c_in = a.groups * CG, c_out = a.groups * MG;
  const rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)gdy, 0, a.B * c_out * a.L_out * 4, 0x00020000);
  for (int item = blockIdx.x * 4 + wave; item < a.n_items; item += gridDim.x * 4) {
    const int per_g = a.csets * a.tiles;
    const int g = item / per_g;
    const int rest = item - g * per_g;
    const int cset = rest / a.tiles, tile = rest - cset * a.tiles;
    const int clip = cset * CPW + sub;
    const int i0 = (tile * LPC + ll) * P;             // the lane's first input position (a multiple of S)
    f32x2 acc[CG / 2][P];
#pragma unroll
    for (int c = 0; c < CG / 2; ++c)
#pragma unroll
      for (int p = 0; p < P; ++p) { acc[c][p].x = 0.f; acc[c][p].y = 0.f; }
    const int q_tile = tile * SPAN + M0;              // dy index of the tile's first staged sample
    const float* wg = gw + (size_t)g * (MG * GK * CG);
    unsigned soff[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int j = lane + 64 * i;
      const int sc = j / ROWF, w = j - sc * ROWF;
      const int q = q_tile + w;
      const int c = cset * CPW + sc;
      const bool ok = j < CPW * ROWF && q >= 0 && q < a.L_out && c < a.B;
      soff[i] = ok ? (unsigned)((c * c_out + g * MG) * a.L_out + q) * 4u : 0x80000000u;
    }
    float st[NLD];
    auto fetch = [&](int oc) __attribute__((always_inline)) {
      const unsigned coff = (unsigned)(oc * a.L_out) * 4u;
#pragma unroll
      for (int i = 0; i < NLD; ++i) st[i] = gload(ry, soff[i] + coff);
    };
    auto publish = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < NLD; ++i) {
        const int j = lane + 64 * i;
        if (j < CPW * ROWF) ys[wave][buf][j] = st[i];
      }
    };
    fetch(0);
    publish(0);
#pragma unroll 1
    for (int oc = 0; oc < MG; ++oc) {
      if (oc + 1 < MG) fetch(oc + 1);
      f32x2 win[WINP / 2];
      const float* yw = &ys[wave][oc & 1][sub * ROWF + ll * (P / S)];
#pragma unroll
      for (int j = 0; j < WINP; j += 4) {
        const f32x4 q = *reinterpret_cast<const f32x4*>(yw + j);
        win[j / 2].x = q.x; win[j / 2].y = q.y; win[j / 2 + 1].x = q.z; win[j / 2 + 1].y = q.w;
      }
      const float* wrow = wg + oc * (GK * CG);
      f32x8 wq[2][4];
#pragma unroll
      for (int h = 0; h < 4; ++h) sload(wq[0][h], wrow + 8 * h);
      all_taps_bwd<CG, S, P, WINP / 2>(acc, win, wq, wrow, std::make_integer_sequence<int, (GK + 32 / CG - 1) / (32 / CG)>{});
      if (oc + 1 < MG) publish((oc + 1) & 1);
    }
    // ---- dx = mask(x) * acc: P consecutive positions per (lane, input channel); 16-byte accesses when the rows are
    // 16-byte aligned (L_in a multiple of 4: a lane's first position is a multiple of P)
    if (clip < a.B) {
      const size_t rowb = ((size_t)clip * c_in + (size_t)g * CG) * a.L_in;
      const bool vec = (a.L_in & 3) == 0 && ((((size_t)gdx) | ((size_t)gmask) | ((size_t)gres)) & 15) == 0;
#pragma unroll
      for (int c = 0; c < CG / 2; ++c) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const size_t rb = rowb + (size_t)(2 * c + h) * a.L_in;
          if (vec) {
#pragma unroll
            for (int p = 0; p < P; p += 4) {
              if (i0 + p < a.L_in) {                  // whole quads: L_in is a multiple of 4
                f32x4 v;
                v.x = h ? acc[c][p].y : acc[c][p].x;
                v.y = h ? acc[c][p + 1].y : acc[c][p + 1].x;
                v.z = h ? acc[c][p + 2].y : acc[c][p + 2].x;
                v.w = h ? acc[c][p + 3].y : acc[c][p + 3].x;
                if (gmask) {
                  const f32x4 m = *reinterpret_cast<const f32x4*>(gmask + rb + i0 + p);
                  v.x *= m.x > 0.f ? 1.f : a.slope; v.y *= m.y > 0.f ? 1.f : a.slope;
                  v.z *= m.z > 0.f ? 1.f : a.slope; v.w *= m.w > 0.f ? 1.f : a.slope;
                }
                if (gres) {
                  const f32x4 r = *reinterpret_cast<const f32x4*>(gres + rb + i0 + p);
                  v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
                }
                *reinterpret_cast<f32x4*>(gdx + rb + i0 + p) = v;
              }
            }
          } else {
#pragma unroll
            for (int p = 0; p < P; ++p) {
              if (i0 + p < a.L_in) {
                float v = h ? acc[c][p].y : acc[c][p].x;
                if (gmask) v *= gmask[rb + i0 + p] > 0.f ? 1.f : a.slope;
                if (gres) v += gres[rb + i0 + p];
                gdx[rb + i0 + p] = v;
              }
            }
          }
        }
      }
    }
  }
}

template <int MG, int CG, int S, int P, int LPC>
int launch_bwd_lpc(const RtgGconvDesc* d, const float* dy, const float* w, const float* mask, const float* res, float* dx,
                   const GcArgs& a0,
                   hipStream_t s) {
  GcArgs a = a0;
  a.tiles = rtg_ceil_div(d->L_in, LPC * P);
  a.csets = rtg_ceil_div(d->B, 64 / LPC);
  const long long items = (long long)d->groups * a.csets * a.tiles;
  if (items > (1ll << 30)) return RTG_ERANGE;
  a.n_items = (int)items;
  int blocks = rtg_ceil_div(items, 4);
  if (blocks > 2048) blocks = 2048;
  RTG_KLAUNCH((gconv_bwd_kernel<MG, CG, S, P, LPC>), dim3(blocks), dim3(RTG_THREADS), 0, s, dy, w, mask, res, dx, a);
  return rtg_launch_status();
}

template <int MG, int CG, int S, int P>
int launch_bwd(const RtgGconvDesc* d, const float* dy, const float* w, const float* mask, const float* res, float* dx,
               const GcArgs& a,
               hipStream_t s) {
  const int need = rtg_ceil_div(d->L_in, P);
  if (need > 32) return launch_bwd_lpc<MG, CG, S, P, 64>(d, dy, w, mask, res, dx, a, s);
  if (need > 16) return launch_bwd_lpc<MG, CG, S, P, 32>(d, dy, w, mask, res, dx, a, s);
  if (need > 8) return launch_bwd_lpc<MG, CG, S, P, 16>(d, dy, w, mask, res, dx, a, s);
  return launch_bwd_lpc<MG, CG, S, P, 8>(d, dy, w, mask, res, dx, a, s);
}

// ---------------------------------------------------------------------------------------------------------------
// backward-weight: dW[oc][ci][t] = sum over (clip, q) of dy[oc][q] * act(x[ci][q*S + t - PAD]), bias: sum of dy.
// Here the OUTPUTS are spread over the lanes: a lane owns COLS of a group's CG*41 + 1 (input channel, tap) columns (the
// last one is the bias: x = 1) for 8 output channels, and walks positions; dy is uniform over the lanes: SGPR pairs of
// two consecutive positions (one s_buffer_load_dwordx4 per output channel and 4 positions, range-checked: the row tail
// reads as 0), x pairs of the same two positions come from the wave's staged LDS rows (two reads S apart per column).
// One packed FMA adds both positions into an (even, odd) accumulator pair, summed at the end.  No cross-lane reduction;
// a wave handles one (group, output-channel half, split) and writes one split partial in the bank's layout
// ([split][rows][CG*41] + [rows] bias), reduced in fixed order by rtg_weightnorm_backward like every other partial.
// ---------------------------------------------------------------------------------------------------------------
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct GwArgs {
  int B, groups, L_in, L_out;
  int W, bpc, n_blocks, per;                // splits, position blocks per clip, blocks in all, blocks per split
  long long part_stride;
  float slope;
};

__device__ __forceinline__ void sbload4(f32x4& w, u32x4 rsrc, unsigned off) {
  asm volatile("s_buffer_load_dwordx4 %0, %1, %2" : "=&s"(w) : "s"(rsrc), "s"(off));
}
__device__ __forceinline__ void swait4(f32x4& w) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(w)); }
// acc.{lo,hi} += dy.{lo,hi} * x.{lo,hi}
__device__ __forceinline__ void pkfma2(f32x2& acc, f32x2 w, f32x2 x) {
  asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "s"(w), "v"(x));
}

template <int MG, int CG, int S>
__global__ __launch_bounds__(RTG_THREADS) void gconv_wgrad_kernel(const float* __restrict__ gx, const float* __restrict__ gdy,
                                                                  float* __restrict__ gpart, const GwArgs a) {
  constexpr int OCW = 8, HALVES = MG / OCW;
  constexpr int N = CG * GK, COLS = (N + 1 + 63) / 64;
  constexpr int PB = 32;                                    // positions per staged block
  constexpr int ROWW = (((PB - 1) * S + GK) + 3) & ~3;      // staged samples per input channel
  constexpr int NLD = (CG * ROWW + 63) / 64;
  constexpr int ONES = CG * ROWW, ZEROS = ONES + ROWW;
  __shared__ __attribute__((aligned(16))) float xs[4][ZEROS + ROWW];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wi = blockIdx.x * 4 + wave;
  if (wi >= a.groups * HALVES * a.W) return;                // (no block barriers below: every wave is on its own)
  const int split = wi % a.W;
  const int gh = wi / a.W;
  const int half = gh % HALVES, g = gh / HALVES;
  const int c_in = a.groups * CG, c_out = a.groups * MG;
  float* xw = xs[wave];
  for (int k = lane; k < ROWW; k += 64) { xw[ONES + k] = 1.f; xw[ZEROS + k] = 0.f; }
  int colbase[COLS];
#pragma unroll
  for (int j = 0; j < COLS; ++j) {
    const int n = lane + 64 * j;
    const int ci = n / GK, t = n - ci * GK;
    colbase[j] = n < N ? ci * ROWW + t : (n == N ? ONES : ZEROS);
  }
  f32x2 acc[OCW][COLS];
#pragma unroll
  for (int o = 0; o < OCW; ++o)
#pragma unroll
    for (int j = 0; j < COLS; ++j) { acc[o][j].x = 0.f; acc[o][j].y = 0.f; }
  const rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)gx, 0, a.B * c_in * a.L_in * 4, 0x00020000);
  const bool act = a.slope != 1.f;
  const int blk0 = split * a.per;
  int blk1 = blk0 + a.per;
  if (blk1 > a.n_blocks) blk1 = a.n_blocks;
  float st[NLD];
  auto fetch = [&](int blk) __attribute__((always_inline)) {
    const int b = blk / a.bpc, q0 = (blk - b * a.bpc) * PB;
    const int e0 = q0 * S - GPAD;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int e = lane + 64 * i;
      const int ci = e / ROWW, k = e - ci * ROWW;
      const int idx = e0 + k;
      const bool ok = e < CG * ROWW && idx >= 0 && idx < a.L_in;
      st[i] = gload(rx, ok ? (unsigned)((b * c_in + g * CG + ci) * a.L_in + idx) * 4u : 0x80000000u);
    }
  };
  auto publish = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int e = lane + 64 * i;
      float v = st[i];
      if (act) v = v > 0.f ? v : v * a.slope;
      if (e < CG * ROWW) xw[e] = v;
    }
  };
  if (blk0 < blk1) fetch(blk0);
#pragma unroll 1
  for (int blk = blk0; blk < blk1; ++blk) {
    publish();
    if (blk + 1 < blk1) fetch(blk + 1);               // in flight during this block's FMAs
    const int b = blk / a.bpc, q0 = (blk - b * a.bpc) * PB;
    // dy rows of this wave's 8 output channels: one range-checked descriptor per row (built on the scalar unit)
    const unsigned long long row0 = (unsigned long long)gdy +
                                    ((unsigned long long)(b * c_out + g * MG + half * OCW) * a.L_out) * 4ull;
    auto rsrc_of = [&](int o) __attribute__((always_inline)) {
      const unsigned long long p = row0 + (unsigned long long)o * a.L_out * 4ull;
      u32x4 r;
      r.x = (unsigned)p; r.y = (unsigned)(p >> 32) & 0xffffu; r.z = (unsigned)a.L_out * 4u; r.w = 0x00020000u;
      return r;
    };
    // sub-blocks of 4 positions: the dy of sub-block sb + 1 is requested right after the wait for sub-block sb and has
    // its 96 packed FMAs to land.  (Requesting the x pairs one sub-block ahead as well costs 24 registers and a wave per
    // SIMD: slower, 0.34 -> 0.40 ms on the largest layer.)
    f32x4 dq[2][OCW];
#pragma unroll
    for (int o = 0; o < OCW; ++o) sbload4(dq[0][o], rsrc_of(o), (unsigned)q0 * 4u);
#pragma unroll
    for (int sb = 0; sb < PB / 4; ++sb) {
      // x pairs of the sub-block's two position pairs, for the lane's columns
      f32x2 xv[COLS][2];
#pragma unroll
      for (int j = 0; j < COLS; ++j)
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
          const float* xp = xw + colbase[j] + (sb * 4 + pp * 2) * S;
          xv[j][pp].x = xp[0];
          xv[j][pp].y = xp[S];
        }
#pragma unroll
      for (int o = 0; o < OCW; ++o) swait4(dq[sb % 2][o]);
      if (sb + 1 < PB / 4) {
#pragma unroll
        for (int o = 0; o < OCW; ++o) sbload4(dq[(sb + 1) % 2][o], rsrc_of(o), (unsigned)(q0 + (sb + 1) * 4) * 4u);
      }
#pragma unroll
      for (int pp = 0; pp < 2; ++pp)
#pragma unroll
        for (int o = 0; o < OCW; ++o) {
          const f32x4 d4 = dq[sb % 2][o];
          f32x2 d2;
          d2.x = pp ? d4.z : d4.x; d2.y = pp ? d4.w : d4.y;
#pragma unroll
          for (int j = 0; j < COLS; ++j) pkfma2(acc[o][j], d2, xv[j][pp]);
        }
    }
  }
  // ---- this wave's split partial
  float* wpart = gpart + (size_t)split * a.part_stride;
  float* bpart = wpart + (size_t)c_out * N;
#pragma unroll
  for (int o = 0; o < OCW; ++o) {
    const int row = g * MG + half * OCW + o;
#pragma unroll
    for (int j = 0; j < COLS; ++j) {
      const int n = lane + 64 * j;
      const float v = acc[o][j].x + acc[o][j].y;
      if (n < N) wpart[(size_t)row * N + n] = v;
      else if (n == N) bpart[row] = v;
    }
  }
}

int gconv_wgrad_kind(const RtgWgradDesc* d) {
  if (!d || d->K != GK || d->pad != GPAD || d->dil != 1 || d->C2 != 0 || d->groups < 2 || d->bf16) return 0;
  if (d->h_k > 1 || d->h_n > 1 || d->gy_mode != RTG_PRE_NONE) return 0;
  if (d->pre_mode != RTG_PRE_NONE && d->pre_mode != RTG_PRE_LRELU) return 0;
  if (d->C1 != d->groups * d->Cg || d->dy_L != d->Q) return 0;
  if (d->Q != (d->L_in + 2 * GPAD - (GK - 1) - 1) / d->stride + 1) return 0;
  if ((long long)d->B * d->groups * (d->Cg > d->Mg ? d->Cg : d->Mg) * (d->L_in > d->Q ? d->L_in : d->Q) * 4 >= (1ll << 31))
    return 0;
  if (d->Mg == 16 && d->Cg == 8 && d->stride == 2) return 1;
  if (d->Mg == 16 && d->Cg == 4 && d->stride == 4) return 2;
  if (d->Mg == 8 && d->Cg == 8 && d->stride == 4) return 3;
  return 0;
}

template <int MG, int CG, int S>
int launch_wgrad(const RtgWgradDesc* d, const float* x, const float* dy, float* part, hipStream_t s) {
  GwArgs a;
  a.B = d->B; a.groups = d->groups; a.L_in = d->L_in; a.L_out = d->Q;
  a.W = d->splits;
  a.bpc = rtg_ceil_div(d->Q, 32);
  a.n_blocks = d->B * a.bpc;
  a.per = rtg_ceil_div(a.n_blocks, a.W);
  a.part_stride = d->part_stride;
  a.slope = d->pre_mode == RTG_PRE_LRELU ? d->pre_slope : 1.f;
  const int waves = d->groups * (MG / 8) * a.W;
  RTG_KLAUNCH((gconv_wgrad_kernel<MG, CG, S>), dim3(rtg_ceil_div(waves, 4)), dim3(RTG_THREADS), 0, s, x, dy, part, a);
  return rtg_launch_status();
}

// instance serving the problem: 1 = (16, 8, s2), 2 = (16, 4, s4), 3 = (8, 8, s4); 0 = none
int gconv_kind(const RtgGconvDesc* d) {
  if (!d || d->K != GK || d->B < 1 || d->groups < 1 || d->L_in < 1 || d->L_out < 1 || d->pad != GPAD) return 0;
  if ((long long)d->B * d->groups * (d->Cg > d->Mg ? d->Cg : d->Mg) * (d->L_in > d->L_out ? d->L_in : d->L_out) * 4 >=
      (1ll << 31))
    return 0;
  if (d->L_out != (d->L_in + 2 * d->pad - (GK - 1) - 1) / d->stride + 1) return 0;
  if (d->Mg == 16 && d->Cg == 8 && d->stride == 2) return 1;
  if (d->Mg == 16 && d->Cg == 4 && d->stride == 4) return 2;
  if (d->Mg == 8 && d->Cg == 8 && d->stride == 4) return 3;
  return 0;
}

}  // namespace

extern "C" int rtg_gconv_ok(const RtgGconvDesc* d) { return gconv_kind(d) > 0 ? 1 : 0; }

extern "C" long long rtg_gconv_workspace(const RtgGconvDesc* d) {
  // + 128: the last block of taps is requested whole (up to 3 taps past the end of the last group's weights)
  return gconv_kind(d) ? (long long)d->groups * d->Mg * d->Cg * d->K + 128 : 0;
}

extern "C" int rtg_gconv_prepare(const RtgGconvDesc* d, const float* v, const float* scale, float* w, void* stream) {
  if (!d || !v || !scale || !w) return RTG_ENULL;
  if (!gconv_kind(d)) return RTG_EINVAL;
  const int total = d->groups * d->Mg * d->Cg * d->K;
  RTG_KLAUNCH(gconv_prep_kernel, dim3(rtg_ceil_div(total, RTG_THREADS)), dim3(RTG_THREADS), 0, (hipStream_t)stream, v, scale, w,
              d->Mg, d->Cg * d->K, total);
  return rtg_launch_status();
}

extern "C" int rtg_gconv_prepare_bwd(const RtgGconvDesc* d, const float* v, const float* scale, float* w, void* stream) {
  if (!d || !v || !scale || !w) return RTG_ENULL;
  if (!gconv_kind(d)) return RTG_EINVAL;
  const int total = d->groups * d->Mg * d->Cg * d->K;
  RTG_KLAUNCH(gconv_prep_bwd_kernel, dim3(rtg_ceil_div(total, RTG_THREADS)), dim3(RTG_THREADS), 0, (hipStream_t)stream, v, scale,
              w, d->Cg, d->K, total);
  return rtg_launch_status();
}

extern "C" int rtg_gconv_backward_data(const RtgGconvDesc* d, const float* dy, const float* w, const float* mask,
                                       const float* res, float* dx, void* stream) {
  if (!d || !dy || !w || !dx) return RTG_ENULL;
  const int kind = gconv_kind(d);
  if (!kind) return RTG_EINVAL;
  GcArgs a;
  a.B = d->B; a.groups = d->groups; a.L_in = d->L_in; a.L_out = d->L_out; a.pad = d->pad;
  a.slope = d->pre_slope;
  a.tiles = 0; a.csets = 0; a.n_items = 0;
  hipStream_t s = (hipStream_t)stream;
  if (kind == 1) return launch_bwd<16, 8, 2, 8>(d, dy, w, mask, res, dx, a, s);
  if (kind == 2) return launch_bwd<16, 4, 4, 16>(d, dy, w, mask, res, dx, a, s);
  return launch_bwd<8, 8, 4, 8>(d, dy, w, mask, res, dx, a, s);
}

extern "C" int rtg_gconv_forward(const RtgGconvDesc* d, const float* x, const float* w, const float* bias, float* out,
                                 void* stream) {
  if (!d || !x || !w || !out) return RTG_ENULL;
  const int kind = gconv_kind(d);
  if (!kind) return RTG_EINVAL;
  GcArgs a;
  a.B = d->B; a.groups = d->groups; a.L_in = d->L_in; a.L_out = d->L_out; a.pad = d->pad;
  a.slope = d->pre_slope;
  a.tiles = 0; a.csets = 0; a.n_items = 0;
  hipStream_t s = (hipStream_t)stream;
  if (kind == 1) return launch_fwd<16, 8, 2, 4>(d, x, w, bias, out, a, s);
  if (kind == 2) return launch_fwd<16, 4, 4, 4>(d, x, w, bias, out, a, s);
  return launch_fwd<8, 8, 4, 4>(d, x, w, bias, out, a, s);
}

// ---- hooks of rtg_wgrad.hip (shape code 9 of RtgWgradDesc.shape_cfg)
int rtg_gconv_wgrad_ok(const RtgWgradDesc* d) {
  return gconv_wgrad_kind(d) > 0 ? 1 : 0;
}

int rtg_gconv_wgrad_splits(const RtgWgradDesc* d) {
  const int kind = gconv_wgrad_kind(d);
  if (!kind) return RTG_EINVAL;
  // a wave per (group, 8 output channels, split): about 3000 waves in all (three per SIMD), at least 8 position blocks each
  const int gh = d->groups * (d->Mg / 8);
  const int n_blocks = d->B * rtg_ceil_div(d->Q, 32);
  int w = rtg_ceil_div(3072, gh);
  const int w_max = n_blocks / 8 > 0 ? n_blocks / 8 : 1;
  if (w > w_max) w = w_max;
  if (w > 512) w = 512;
  return w < 1 ? 1 : w;
}

int rtg_gconv_wgrad_launch(const RtgWgradDesc* d, const float* x, const float* dy, float* part, hipStream_t s) {
  const int kind = gconv_wgrad_kind(d);
  if (!kind) return RTG_EINVAL;
  if (d->splits != rtg_gconv_wgrad_splits(d)) return RTG_EINVAL;
  if (kind == 1) return launch_wgrad<16, 8, 2>(d, x, dy, part, s);
  if (kind == 2) return launch_wgrad<16, 4, 4>(d, x, dy, part, s);
  return launch_wgrad<8, 8, 4>(d, x, dy, part, s);
}
