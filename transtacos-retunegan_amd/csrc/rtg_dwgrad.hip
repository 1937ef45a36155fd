// rtg_dwgrad.hip — weight / bias gradients of the dense discriminator layers (the layers rtg_dconv.hip serves forward:
// DiscriminatorP convs.1-4, DiscriminatorS convs.5; discrminator.py:44,155-163) with 16-byte operand fragments.
//
//   dW[m][c][t] = sum over (clip, q) of gy[clip, m, q] * lrelu(x[clip, c, q * S + t - pad]),   db[m] = sum gy[clip, m, q]
//
// GEMM view: rows = output channels, columns = (input channel, tap) pairs of a 16-channel chunk (80 columns at k = 5), the
// reduction runs over the dense (clip, q) sequence n = clip * Q + q in tiles of 64.  Both operand tiles are staged in LDS
// in the order the matrix cores read them: [16 reductions n][row or column][n % 16], so that lane (kgrp, r) of a
// v_mfma_f32_16x16x4_f32 fetches its operands of FOUR k-steps (n = 16 g + 4 kgrp + 0..3) with one aligned ds_read_b128
// from either image — the column image is the im2col of the tile ([tap] shifted copies of the input rows), which is what
// makes the shifted reads of the taps aligned.  The general kernel (rtg_wgrad_kernel.h) reads one float per matrix
// instruction and operand from a [row][64] tile and the raw patch.
// A block (8 waves, one 16-row tile each, all five column tiles) owns 128 rows x one channel chunk x one split of the
// reduction; the tiles of its split are double buffered in LDS (global -> registers during the multiplications, registers
// -> LDS behind them, ONE barrier per tile).  Chunk 0's blocks also accumulate the bias gradient (a ones operand).
// Exposed as shape code 10 of RtgWgradDesc.shape_cfg: the tuner times it against the general shapes.  Its summation
// order differs from theirs (rounding level), results are reproducible run to run (no atomics; splits are summed in fixed
// order by rtg_weightnorm_backward).
#include <type_traits>

#include "rtg_common.h"

namespace {

using rsrc_t = __amdgpu_buffer_rsrc_t;
#define DW_OOB 0x80000000u
constexpr int kCch = 16, kK = 5, kCols = kCch * kK, kNCT = kCols / 16, kTT = 64, kNG = kTT / 16;
constexpr int kBF = kNG * kCols * 16;                                // floats per buffer of the column image

struct WArgs {
  const float *x, *dy;
  float* part;
  int B, Cg, L_in, Mg, Q, dy_L, pad;
  float xslope, gy_scale, inv_Q;
  int splits;
  long long part_stride;
  int n_red, n_tiles, n_mb, n_cch, per_split, n_items, per_xcd;
  int x_bytes, dy_bytes;
};

__device__ __forceinline__ float dw_load(rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}

// S: stride; kWB waves of one 16-row tile each (8: 128 rows, one block per CU; 4: 64 rows, two blocks per CU whose barrier
// and staging phases overlap each other's multiplications, at twice the input staging per row)
template <int S, int kWB>
__global__ __launch_bounds__(kWB * 64, 2) void dwgrad_kernel(const WArgs a) {
  constexpr int kRows = kWB * 16;
  constexpr int kAF = kNG * kRows * 16;                                // floats per buffer of the row image
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* const la = lds;                       // [2][kNG][kRows][16]
  float* const lb = lds + 2 * kAF;             // [2][kNG][kCols][16]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // block -> (split, row block, channel chunk): each XCD walks a contiguous range of items; the chunks of one (split,
  // row block) are neighbours (same gy tiles), the row blocks of one split next (same input tiles)
  const int item = (int)(blockIdx.x & 7u) * a.per_xcd + (int)(blockIdx.x >> 3);
  if (item >= a.n_items) return;
  const int split = item / a.per_split;
  int rem = item - split * a.per_split;
  const int cch = rem % a.n_cch, mb = rem / a.n_cch;
  const int m0 = mb * kRows, c0 = cch * kCch;

  const rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
  const rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, a.dy_bytes, 0x00020000);
  const unsigned rowb_d = (unsigned)a.dy_L * 4u, rowb_x = (unsigned)a.L_in * 4u;

  // ---- staging: the lane is reduction index n of the tile; the wave fetches rows wave, wave + 8, ... of gy and the K
  // shifted copies of channels wave, wave + 8 of the chunk
  float sa[kRows / kWB], sb[kCch / kWB][kK];
  auto stage_load = [&](int tile) __attribute__((always_inline)) {
    const int n = tile * kTT + lane;
    int clip = (int)((float)n * a.inv_Q);
    int q = n - clip * a.Q;
    if (q < 0) { --clip; q += a.Q; }
    else if (q >= a.Q) { ++clip; q -= a.Q; }
    const bool valid = n < a.n_red;
    const unsigned va = valid ? ((unsigned)clip * (unsigned)a.Mg * (unsigned)a.dy_L + (unsigned)q) * 4u : DW_OOB;
#pragma unroll
    for (int i = 0; i < kRows / kWB; ++i) sa[i] = dw_load(rd, va, (unsigned)(m0 + wave + kWB * i) * rowb_d);
    const unsigned xclip = (unsigned)clip * (unsigned)a.Cg * (unsigned)a.L_in;
#pragma unroll
    for (int t = 0; t < kK; ++t) {
      const int pos = q * S + t - a.pad;
      const unsigned vb = (valid && pos >= 0 && pos < a.L_in) ? (xclip + (unsigned)pos) * 4u : DW_OOB;
#pragma unroll
      for (int j = 0; j < kCch / kWB; ++j) sb[j][t] = dw_load(rx, vb, (unsigned)(c0 + wave + kWB * j) * rowb_x);
    }
  };
  auto stage_write = [&](int buf) __attribute__((always_inline)) {
    float* pa = la + buf * kAF + (lane >> 4) * kRows * 16 + (lane & 15) + wave * 16;
#pragma unroll
    for (int i = 0; i < kRows / kWB; ++i) {
      float v = sa[i];
      asm volatile("" : "+v"(v));                   // keep the consumption (and its wait) here, below the multiplications
      pa[i * kWB * 16] = v * a.gy_scale;
    }
    float* pb = lb + buf * kBF + (lane >> 4) * kCols * 16 + (lane & 15) + wave * kK * 16;
#pragma unroll
    for (int j = 0; j < kCch / kWB; ++j)
#pragma unroll
      for (int t = 0; t < kK; ++t) {
        float v = sb[j][t];
        asm volatile("" : "+v"(v));
        pb[(j * kWB * kK + t) * 16] = v > 0.f ? v : v * a.xslope;
      }
  };

  // ---- operands of this wave: row tile `wave`, every column tile
  const int r16 = lane & 15, kgrp = lane >> 4;
  const int aoff = (wave * 16 + r16) * 16 + kgrp * 4;
  const int boff = r16 * 16 + kgrp * 4;
  f32x4 acc[kNCT], accb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < kNCT; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool with_bias = cch == 0;

  struct Frag {
    f32x4 a, b[kNCT];
  };
  int tile = split;
  if (tile < a.n_tiles) {
    stage_load(tile);
    stage_write(0);
  }
  __syncthreads();
  // BIAS (chunk 0's blocks): one more matrix instruction per k-step against a ones operand — two copies of the loop, so
  // that the other blocks carry no branch inside the multiplications
  auto run = [&](auto bias_tag) __attribute__((always_inline)) {
    constexpr bool BIAS = decltype(bias_tag)::value;
    int cur = 0;
    for (; tile < a.n_tiles; tile += a.splits) {
      const int nxt = tile + a.splits;
      // (unconditional: past the last tile every offset is out of range, the loads return zeros nobody writes — a branch
      // around them would make the compiler wait for ALL loads at the join)
#ifndef RTG_EXP_DW_NOLOAD
      stage_load(nxt < a.n_tiles ? nxt : (1 << 24));
#endif
      const float* pa = la + cur * kAF + aoff;
      const float* pb = lb + cur * kBF + boff;
      auto fetch = [&](Frag& f, int g) __attribute__((always_inline)) {
        f.a = *reinterpret_cast<const f32x4*>(pa + g * kRows * 16);
#pragma unroll
        for (int j = 0; j < kNCT; ++j) f.b[j] = *reinterpret_cast<const f32x4*>(pb + (g * kCols + j * 16) * 16);
      };
      auto mma = [&](const Frag& f) __attribute__((always_inline)) {
#pragma unroll
        for (int kq = 0; kq < 4; ++kq) {
#pragma unroll
          for (int j = 0; j < kNCT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[kq], f.b[j][kq], acc[j], 0, 0, 0);
          if constexpr (BIAS) accb = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[kq], 1.0f, accb, 0, 0, 0);
        }
      };
      // the fragments of 16-reduction group g + 1 are requested before group g is multiplied
      Frag f0, f1;
      fetch(f0, 0);
#pragma unroll
      for (int g = 0; g < kNG; ++g) {
        Frag& fc = (g & 1) ? f1 : f0;
        Frag& fn = (g & 1) ? f0 : f1;
        if (g + 1 < kNG) fetch(fn, g + 1);
        __builtin_amdgcn_sched_barrier(0);
        mma(fc);
        __builtin_amdgcn_sched_barrier(0);
      }
#ifndef RTG_EXP_DW_NOWRITE
      if (nxt < a.n_tiles) stage_write(cur ^ 1);
#endif
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      cur ^= 1;
    }
  };
  if (with_bias) run(std::true_type{});
  else run(std::false_type{});

  // ---- this split's partial: [rows][Cg * K] then the bias partials
  float* wpart = a.part + (size_t)split * a.part_stride;
  const int ck = a.Cg * kK;
#pragma unroll
  for (int j = 0; j < kNCT; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m0 + wave * 16 + kgrp * 4 + r;
      wpart[(size_t)m * ck + c0 * kK + j * 16 + r16] = acc[j][r];
    }
  if (with_bias && r16 == 0) {
    float* bpart = wpart + (size_t)a.Mg * ck;
#pragma unroll
    for (int r = 0; r < 4; ++r) bpart[m0 + wave * 16 + kgrp * 4 + r] = accb[r];
  }
}

constexpr int kRowsOf[2] = {128, 64};                                  // shape code 10, 11

bool eligible(const RtgWgradDesc* d, int kRows = 128) {
  if (d->groups != 1 || d->C2 != 0 || d->h_k > 1 || d->h_n > 1 || d->bf16) return false;
  if (d->K != kK || d->dil != 1 || (d->stride != 1 && d->stride != 3)) return false;
  if (d->Cg != d->C1 || d->Cg % kCch != 0 || d->Mg % kRows != 0) return false;
  if (d->gy_mode != RTG_PRE_NONE || (d->pre_mode != RTG_PRE_NONE && d->pre_mode != RTG_PRE_LRELU)) return false;
  if (d->Q < 1 || d->Q > d->dy_L) return false;
  const long long n = (long long)d->B * d->Q;
  if (n >= (1ll << 23)) return false;                                         // float-reciprocal division of n by Q
  if ((long long)d->B * d->C1 * d->L_in * 4 >= (1ll << 31) || (long long)d->B * d->Mg * d->dy_L * 4 >= (1ll << 31)) return false;
  return true;
}

}  // namespace

// variant 0: 128-row blocks (shape code 10), 1: 64-row blocks (11)
int rtg_dwgrad_ok(const RtgWgradDesc* d, int variant) { return (variant == 0 || variant == 1) && eligible(d, kRowsOf[variant]) ? 1 : 0; }

// split count: one block per CU (104 KB of LDS, 8 waves); a launch lasts (rounds of the chip) x (tiles per block) tile times
// plus the write and the fixed-order read-back of one partial per split
int rtg_dwgrad_splits(const RtgWgradDesc* d, int variant) {
  if (!rtg_dwgrad_ok(d, variant)) return RTG_EINVAL;
  const int kRows = kRowsOf[variant];
  const long long base = (long long)(d->Mg / kRows) * (d->Cg / kCch);
  const long long tiles = ((long long)d->B * d->Q + kTT - 1) / kTT;
  const double t_tile = variant == 0 ? 2.4 : 1.3, t_fixed = 6.0;
  const long long slots = variant == 0 ? 256 : 512;
  const double t_flush = (double)d->Mg * ((double)d->Cg * d->K + 1) * 8.0 / 3.0e6;
  double best = 1e30;
  long long best_s = 1;
  const long long s_max = tiles < 512 ? tiles : 512;
  for (long long s = 1; s <= s_max; ++s) {
    const long long rounds = (base * s + slots - 1) / slots;
    const double t = (double)rounds * ((double)((tiles + s - 1) / s) * t_tile + t_fixed) + (double)s * t_flush;
    if (t < best) { best = t; best_s = s; }
  }
  return (int)best_s;
}

int rtg_dwgrad_launch(const RtgWgradDesc* d, int variant, const float* x, const float* dy, float* part, hipStream_t s) {
  if (!rtg_dwgrad_ok(d, variant)) return RTG_EINVAL;
  const int kRows = kRowsOf[variant], kWB = kRows / 16;
  if (!x || !dy || !part) return RTG_ENULL;
  WArgs a;
  a.x = x; a.dy = dy; a.part = part;
  a.B = d->B; a.Cg = d->Cg; a.L_in = d->L_in; a.Mg = d->Mg; a.Q = d->Q; a.dy_L = d->dy_L; a.pad = d->pad;
  a.xslope = d->pre_mode == RTG_PRE_LRELU ? d->pre_slope : 1.f;
  a.gy_scale = d->gy_scale;
  a.inv_Q = 1.0f / (float)d->Q;
  a.splits = d->splits; a.part_stride = d->part_stride;
  a.n_red = d->B * d->Q;
  a.n_tiles = rtg_ceil_div(a.n_red, kTT);
  a.n_mb = d->Mg / kRows; a.n_cch = d->Cg / kCch;
  a.per_split = a.n_mb * a.n_cch;
  const long long n_items = (long long)a.per_split * d->splits;
  if (n_items > (1ll << 28)) return RTG_ERANGE;
  a.n_items = (int)n_items;
  a.per_xcd = (int)((n_items + 7) / 8);
  a.x_bytes = d->B * d->C1 * d->L_in * 4;
  a.dy_bytes = d->B * d->Mg * d->dy_L * 4;
  const size_t lds_bytes = (size_t)2 * (kNG * kRows * 16 + kBF) * sizeof(float);
  static bool attr_set[4] = {false, false, false, false};
  auto k = variant == 0 ? (d->stride == 1 ? dwgrad_kernel<1, 8> : dwgrad_kernel<3, 8>)
                        : (d->stride == 1 ? dwgrad_kernel<1, 4> : dwgrad_kernel<3, 4>);
  bool& set = attr_set[variant * 2 + (d->stride == 1 ? 0 : 1)];
  if (!set) {
    if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return RTG_ERANGE;
    set = true;
  }
  RTG_KLAUNCH(k, dim3((unsigned)(8 * a.per_xcd)), dim3(kWB * 64), lds_bytes, s, a);
  return rtg_launch_status();
}
