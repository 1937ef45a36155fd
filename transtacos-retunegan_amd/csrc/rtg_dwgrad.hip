// rtg_dwgrad.hip — weight / bias gradients of the dense discriminator layers (the layers rtg_dconv.hip serves forward:
// DiscriminatorP convs.1-4, DiscriminatorS convs.5; discrminator.py:44,155-163) with 16-byte operand fragments.
//
//   dW[m][c][t] = sum over (clip, q) of gy[clip, m, q] * lrelu(x[clip, c, q * S + t - pad]),   db[m] = sum gy[clip, m, q]
//
// GEMM view: rows = output channels, columns = (input channel, tap) pairs of NCH 16-channel chunks (80 columns each at
// k = 5), the reduction runs over the dense (clip, q) sequence n = clip * Q + q in tiles of 64.  Lane (kgrp, r) of a
// v_mfma_f32_16x16x4_f32 takes its operands of FOUR k-steps (n = 16 g + 4 kgrp + 0..3) from one 16-byte fetch:
//   rows     gy is read by exactly one wave (waves are stacked along the rows), so it never passes through LDS: a lane
//            loads its four consecutive n of row r straight from global memory (one 16-byte load; a second one from the
//            next clip's row where the four straddle a clip boundary, merged by selects), a tile ahead;
//   columns  the tile's im2col ([tap] shifted copies of the input rows, leaky-relu applied) is staged in LDS as
//            [16 reductions][column][n % 16] — what makes the shifted reads of the taps aligned 16-byte reads — double
//            buffered, ONE barrier per tile.
// The general kernel (rtg_wgrad_kernel.h) reads one float per matrix instruction and operand from a [row][64] gy tile
// and the raw patch, both in LDS.  Measured on the way (ablation builds, tools/dbg/abl_dwgrad.sh; 512 -> 512, k5): with gy
// staged through LDS like the columns the kernel ran 72-81 TFLOP/s, 98-102 without its 26 global loads per wave and
// tile (82 with the loads issued but out of range: the ISSUE of a vector-memory instruction costs a wave ~60 cycles,
// whatever it fetches) and 113-118 without the LDS writes as well — hence 16-byte loads and nothing staged twice.
// A block = kWB waves of one 16-row tile each x NCH channel chunks x one split of the reduction.  Chunk 0's blocks also
// accumulate the bias gradient (a ones operand).  Exposed as shape codes 10.. of RtgWgradDesc.shape_cfg; the tuner times
// them against the general shapes.  The summation order differs from theirs (rounding level); results are reproducible
// run to run (no atomics; splits are summed in fixed order by rtg_weightnorm_backward).
#include <type_traits>

#include "rtg_common.h"

namespace {

using rsrc_t = __amdgpu_buffer_rsrc_t;
using u32x4 = unsigned __attribute__((ext_vector_type(4)));
#define DW_OOB 0x80000000u
constexpr int kCch = 16, kTT = 64, kNG = kTT / 16;

struct WArgs {
  const float *x, *dy;
  float* part;
  int B, Cg, L_in, Mg, Q, dy_L, pad;
  float xslope, gy_scale, inv_Q;
  int splits;
  long long part_stride;
  int n_red, n_tiles, n_mb, n_cch, per_split, n_items, per_xcd;
  int x_bytes, dy_bytes;
  // second dimension (RtgWgradDesc.h_*; forward geometry): a clip is an (item, output row) pair, a channel a (channel,
  // kernel row) pair; x is [items, Cg / h_k, h_in, L_in], dy [items, Mg, h_n, dy_L]
  int h_in, h_k, h_stride, h_pad, h_n;
  int bf, io;                   // (host side: which instance)
};

__device__ __forceinline__ float dw_load(rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ f32x4 dw_load4(rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

// S: stride; kWB: waves; RW: 16-row tiles per wave; NCH: channel chunks per block; kK: taps; TWO_D: the Conv2d layers of
// StftDiscriminator (discrminator.py:255-262) along their last axis
//
// BF (RtgWgradDesc.bf16): bf16 operands on v_mfma_f32_16x16x32_bf16, fp32 accumulation: a lane's fragment is 8 consecutive
// reductions — two of the 4-element gy groups, converted when the fragment is built; the column image holds bf16 (a tile is
// two 32-reduction groups of four planes [kgrp][column][8 bf16], the staging writes are 2-byte), the activation is applied
// in fp32 and rounded to nearest even at the LDS write.  Same loads, same loop; an eighth of the matrix instructions.
//
// IO (RtgWgradDesc.io_bf16, round 5: bf16 feature maps in HBM; BF only).  Bit 0, x is bf16 and already activated (what a
// producer with RTG_IO_OUT_BF16 stored): the K taps of a reduction are one 16-byte load (8 positions; fp32: 16 + 4 bytes) and
// go to the column image as they are — no activation, no conversion, a 2-byte LDS write each.  Bit 1, dy is bf16: a lane's
// fragment of 8 consecutive reductions is ONE 16-byte load (a second one behind a clip boundary, merged by bit selects;
// fp32: four loads and eight conversions); rows of at least 8 positions.
using bf16x8 = __bf16 __attribute__((ext_vector_type(8)));
template <int S, int kWB, int NCH, int RW, int kK, bool TWO_D, bool BF, int IO = 0>
__global__ __launch_bounds__(kWB * 64, 2) void dwgrad_kernel(const WArgs a) {
  static_assert(IO == 0 || BF, "bf16 tensors go with bf16 operands");
  constexpr bool XB = (IO & 1) != 0, YB = (IO & 2) != 0;
  constexpr unsigned XES = XB ? 2u : 4u, YES = YB ? 2u : 4u;           // element sizes of x and dy
  constexpr int NSTEP = BF ? 2 : kNG;                                 // matrix k-groups per 64-reduction tile
  constexpr int kCols = kCch * kK, kNCT = kCols / 16;                 // columns / column tiles of one chunk
  // The column image of one chunk and 16-reduction group: four planes [kgrp][column][4 reductions] (element (column, r) at
  // plane r / 4, slot r % 4: the four k-steps of lane group kgrp).  ds_read_b128 serves a wave in 16-lane groups that pair columns 0-3, 12-15 of one plane with
  // columns 4-11 of the next (MI355X_MICROARCH.md, LDS): planes 0/1 and 2/3 a multiple of 256 bytes apart make a group's 16
  // fragments 16 consecutive columns = all 64 banks once ([column][16 reductions] rows of 64 bytes were 2-way conflicts
  // throughout).  The 16 bytes between planes 1 and 2 and the 32 bytes between groups keep the staging writes
  // (ds_write_b32, lane = reduction index, 32 lanes per LDS cycle) at 2-way, which costs them nothing.
  constexpr int kPS = (kCols * 4 + 63) / 64 * 64;                      // floats between planes 0 / 1 and 2 / 3
  constexpr int kGS = 4 * kPS + 8;                                    // floats per 16-reduction group
  constexpr int kBF = NSTEP * kGS;                                    // floats of one chunk's column image (one buffer)
  constexpr int kRows = kWB * RW * 16;
  constexpr int CPW = NCH * kCch / kWB;                               // input channels a wave stages per tile
  static_assert(NCH * kCch % kWB == 0, "channels split evenly over the waves");
  extern __shared__ __attribute__((aligned(16))) float lds[];      // [2][NCH][kNG][plane 4][kCols][4]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // block -> (split, row block, group of NCH channel chunks): each XCD walks a contiguous range of items; the chunk
  // groups of one (split, row block) are neighbours (same gy rows), the row blocks of one split next (same input tiles)
  const int item = (int)(blockIdx.x & 7u) * a.per_xcd + (int)(blockIdx.x >> 3);
  if (item >= a.n_items) return;
  const int split = item / a.per_split;
  int rem = item - split * a.per_split;
  const int cch = rem % a.n_cch, mb = rem / a.n_cch;
  const int m0 = mb * kRows, c0 = cch * NCH * kCch;

  const rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
  const rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, a.dy_bytes, 0x00020000);
  const int r16 = lane & 15, kgrp = lane >> 4;

  auto divq = [&](int n, int& q) __attribute__((always_inline)) {      // n / Q (n < 2^23) through the float reciprocal
    int c = (int)((float)n * a.inv_Q);
    q = n - c * a.Q;
    if (q < 0) { --c; q += a.Q; }
    else if (q >= a.Q) { ++c; q -= a.Q; }
    return c;
  };

  // ---- rows: this lane's four consecutive reductions of row (wave's tile, r16) for each 16-reduction group of a tile
  constexpr int NAG = YB ? NSTEP : kNG;        // row-operand loads per tile: 4-element groups (fp32 dy) or 8-element fragments
  constexpr int AEL = YB ? 8 : 4;
  f32x4 l1[RW][NAG], l2[RW][NAG];              // (bf16 dy: the 16 bytes are 8 elements)
  int arem[NAG], aval[NAG];                    // elements before the clip boundary; valid elements (n < n_red)
  const unsigned arow = (unsigned)(m0 + wave * RW * 16 + r16) * (unsigned)(a.h_n * a.dy_L) * YES;     // (h_n == 1 in 1-D)
  const unsigned atile = 16u * (unsigned)(a.h_n * a.dy_L) * YES;  // bytes between the wave's row tiles
  // byte offset of (clip, row 0, position 0) in dy: clip = item (1-D) or (item, output row)
  auto dy_clip = [&](int clip) __attribute__((always_inline)) {
    if constexpr (TWO_D) {
      const int item = clip / a.h_n, ho = clip - item * a.h_n;
      return ((unsigned)item * (unsigned)a.Mg * (unsigned)a.h_n + (unsigned)ho) * (unsigned)a.dy_L * YES;
    } else {
      return (unsigned)clip * (unsigned)a.Mg * (unsigned)a.dy_L * YES;
    }
  };
  // (ablation hooks, tools/dbg/abl_dwgrad.sh: RTG_EXP_DW_NOLOAD no global loads, _NOWRITE no column-image writes, _NOFETCH no
  // LDS fragment reads, _NOMMA no matrix instructions, _NOEPI no partial stores — wrong results by design.  Round 5,
  // profiles/r05_dwgrad_ablations.txt: fp32 512 -> 512 k5 194 us; matrix instructions alone 135; 156 without the loads; 168
  // without the image writes; 35 without matrix instructions.  Spreading the image writes of the next tile over the second
  // half of the tile's matrix instructions was built on that and measured: neutral where it fits the registers (351 vs 356
  // us, bf16 110.7 vs 110.8), twice as slow where it spills (the fp32 2-chunk and 4-chunk shapes sit at 249-253 registers).
  // So was a phase shift between the two waves of a SIMD (the upper four waves request their input rows two tiles ahead and
  // write them right behind the barrier, the lower four as now): fp32 271 / 414 / 516 us against 268 / 407 / 515 — the write
  // phases meeting is not what the 60 us over the matrix-only loop are; the ~16 load issues per wave and tile at the top of
  // the iteration are the next suspect (156 us without them).)
  auto a_load = [&](int tile) __attribute__((always_inline)) {
#ifdef RTG_EXP_DW_NOLOAD
    for (int g = 0; g < NAG; ++g) {
      arem[g] = aval[g] = AEL;
      for (int i = 0; i < RW; ++i) l1[i][g] = l2[i][g] = f32x4{1.f, 1.f, 1.f, 1.f};
    }
    return;
#endif
#pragma unroll
    for (int g = 0; g < NAG; ++g) {
      // this lane's four consecutive reductions of group g (bf16: halves g & 1 of the 8 reductions of k-group g >> 1); bf16
      // dy: the 8 reductions of k-group g
      const int n0 = tile * kTT + (YB ? g * 32 + kgrp * 8 : (BF ? (g >> 1) * 32 + kgrp * 8 + (g & 1) * 4 : g * 16 + kgrp * 4));
      int q0;
      const int clip = divq(n0, q0);
      const int left = a.n_red - n0;
      aval[g] = left < 0 ? 0 : left;
      arem[g] = a.Q - q0;
      const unsigned o1 = left > 0 ? dy_clip(clip) + arow + (unsigned)q0 * YES : DW_OOB;
      // the part behind a clip boundary: row r of the NEXT clip, element e at position e - arem.  (The very first bytes of
      // the tensor cannot be addressed from before its start: clip + 1 >= 1 keeps this offset positive.)
      const unsigned o2 = (left > arem[g] && arem[g] < AEL) ? dy_clip(clip + 1) + arow - (unsigned)arem[g] * YES : DW_OOB;
#pragma unroll
      for (int i = 0; i < RW; ++i) {
        l1[i][g] = dw_load4(rd, o1, i * atile);
        l2[i][g] = dw_load4(rd, o2, i * atile);
      }
    }
  };
  // bf16 dy: the fragment of k-group g — elements before the clip boundary from the first load, the rest from the second,
  // nothing past the end of the reduction; as bit selects on the packed pairs
  auto a_frag_b = [&](int i, int g) __attribute__((always_inline)) {
    const u32x4 v1 = __builtin_bit_cast(u32x4, l1[i][g]), v2 = __builtin_bit_cast(u32x4, l2[i][g]);
    u32x4 f;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const unsigned m1 = (2 * d < arem[g] ? 0xffffu : 0u) | (2 * d + 1 < arem[g] ? 0xffff0000u : 0u);
      const unsigned mv = (2 * d < aval[g] ? 0xffffu : 0u) | (2 * d + 1 < aval[g] ? 0xffff0000u : 0u);
      f[d] = ((v1[d] & m1) | (v2[d] & ~m1)) & mv;
    }
    return __builtin_bit_cast(bf16x8, f);
  };
  auto a_frag = [&](int i, int g) __attribute__((always_inline)) {
    f32x4 f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float v = e < arem[g] ? l1[i][g][e] : l2[i][g][e];
      f[e] = e < aval[g] ? v : 0.f;
    }
    return f;
  };

  // ---- columns: the lane is reduction index n of the tile; a wave stages the K shifted copies of CPW channels.  The K taps
  // of a reduction are K CONSECUTIVE input positions (q * S - pad + t), so a channel costs one unaligned 16-byte load (taps
  // 0 .. 3) and, for five taps, one 4-byte load — not K loads: issuing the loads is what this kernel waits for.  Taps outside
  // the row read the neighbouring row's elements and are zeroed by a per-lane tap mask when the image is written; an
  // invalid lane / row loads from an out-of-range offset (zeros).  Only the very first elements of the tensor cannot be
  // addressed that way (the 16 bytes would start before the buffer): the load starts at element 0 and the taps are taken
  // `ssh` elements further left (first staged channel of a wave only: every other channel sits at least a row in).
  f32x4 sbv[CPW];
  [[maybe_unused]] float sb4[CPW];
  constexpr int NSH = kWB >= 8 ? 1 : 2;              // channels wave + kWB * j that can be channel 0 / a kernel row of it
  int ssh[NSH];
  unsigned stv = 0;                                  // bit t: tap t of this lane's reduction lies inside the row
  auto b_load = [&](int tile) __attribute__((always_inline)) {
#ifdef RTG_EXP_DW_NOLOAD
    stv = 31u;
    for (int j = 0; j < NSH; ++j) ssh[j] = 0;
    for (int j = 0; j < CPW; ++j) {
      sbv[j] = f32x4{1.f, 1.f, 1.f, 1.f};
      if constexpr (kK == 5 && !XB) sb4[j] = 1.f;
    }
    return;
#endif
    const int n = tile * kTT + lane;
    int q;
    const int clip = divq(n, q);
    const bool valid = n < a.n_red;
    const int p0 = q * S - a.pad;
    unsigned tvm = 0;
#pragma unroll
    for (int t = 0; t < kK; ++t) tvm |= (valid && (unsigned)(p0 + t) < (unsigned)a.L_in) ? (1u << t) : 0u;
    stv = tvm;
    auto issue = [&](int j, int orig, bool ok) __attribute__((always_inline)) {
      int ai = orig;
      if (j < NSH) {
        ai = orig < 0 ? 0 : orig;
        ssh[j < NSH ? j : 0] = ai - orig;
      }
      sbv[j] = dw_load4(rx, ok ? (unsigned)ai * XES : DW_OOB, 0);       // (bf16 x: the 16 bytes are 8 positions, all taps)
      if constexpr (kK == 5 && !XB) sb4[j] = dw_load(rx, (ok && (tvm & 16u)) ? (unsigned)(orig + 4) * 4u : DW_OOB, 0);
    };
    if constexpr (TWO_D) {
      // channel vc = (c, kh): input row ho * h_stride - h_pad + kh of channel c
      const int item = clip / a.h_n, ho = clip - item * a.h_n;
      const int row0 = ho * a.h_stride - a.h_pad;
      const int cin = a.Cg / a.h_k;
      const int xitem = item * cin * a.h_in * a.L_in + p0;
#pragma unroll
      for (int j = 0; j < CPW; ++j) {
        const int vc = c0 + wave + kWB * j;
        const int c = vc / a.h_k, kh = vc - c * a.h_k;
        const int row = row0 + kh;
        issue(j, xitem + (c * a.h_in + row) * a.L_in, valid && (unsigned)row < (unsigned)a.h_in);
      }
    } else {
      const int xclip = clip * a.Cg * a.L_in + p0;
#pragma unroll
      for (int j = 0; j < CPW; ++j) issue(j, xclip + (c0 + wave + kWB * j) * a.L_in, valid);
    }
  };
  // tap t of staged channel j: element t (- ssh) of the 16 bytes, or the fifth load; zero outside the row
  auto tapval = [&](const f32x4& lv, float l4, int j, int t) __attribute__((always_inline)) {
    float v;
    if (t == 4) {
      v = l4;
    } else {
      v = lv[t];
      if (j < NSH) {
        const int sh = ssh[j < NSH ? j : 0];
        const float v1 = t >= 1 ? lv[t >= 1 ? t - 1 : 0] : 0.f, v2 = t >= 2 ? lv[t >= 2 ? t - 2 : 0] : 0.f;
        v = sh == 0 ? v : (sh == 1 ? v1 : v2);
      }
    }
    return (stv >> t) & 1u ? v : 0.f;
  };
  // (the loaded registers are consumed — and waited for — where the image is written, below the multiplications)
  auto taken = [&](int j, f32x4& lv, float& l4) __attribute__((always_inline)) {
    lv = sbv[j];
    asm volatile("" : "+v"(lv));
    l4 = 0.f;
    if constexpr (kK == 5 && !XB) {
      l4 = sb4[j];
      asm volatile("" : "+v"(l4));
    }
  };
  // bf16 x: tap t of staged channel j as the 16 bits the column image takes — element t (- ssh) of the 8 loaded, zero outside
  // the row; the tensor holds activated values (no leaky-relu here)
  auto tapbits = [&](const u32x4& lv, int j, int t) __attribute__((always_inline)) {
    auto half = [&](int e) __attribute__((always_inline)) {
      return e < 0 ? 0u : ((e & 1) ? lv[e >> 1] >> 16 : lv[e >> 1] & 0xffffu);
    };
    unsigned v = half(t);
    if (j < NSH) {
      const int sh = ssh[j < NSH ? j : 0];
      v = sh == 0 ? v : (sh == 1 ? half(t - 1) : half(t - 2));
    }
    return (unsigned short)((stv >> t) & 1u ? v : 0u);
  };
  auto b_write = [&](int buf) __attribute__((always_inline)) {
#ifdef RTG_EXP_DW_NOWRITE
    return;
#endif
    // channel c = wave + kWB * j of the block's NCH * 16: chunk c / 16, column (c % 16) * K + t
    if constexpr (BF) {
      // reduction n = lane: k-group n / 32, plane (n / 8) % 4, element n % 8 of the column's 16 bytes
      const int kg = (lane >> 3) & 3;
      __bf16* pb = reinterpret_cast<__bf16*>(lds + buf * (NCH * kBF) + (lane >> 5) * kGS + kg * kPS + (kg >> 1) * 4) + (lane & 7);
#pragma unroll
      for (int j = 0; j < CPW; ++j) {
        const int c = wave + kWB * j;
        f32x4 lv;
        float l4;
        taken(j, lv, l4);
#pragma unroll
        for (int t = 0; t < kK; ++t) {
          if constexpr (XB) {
            reinterpret_cast<unsigned short*>(pb)[((c >> 4) * kBF + ((c & 15) * kK + t) * 4) * 2] =
                tapbits(__builtin_bit_cast(u32x4, lv), j, t);
          } else {
            float v = tapval(lv, l4, j, t);
            pb[((c >> 4) * kBF + ((c & 15) * kK + t) * 4) * 2] = (__bf16)(v > 0.f ? v : v * a.xslope);
          }
        }
      }
    } else {
    const int wr = lane & 15;
    float* pb = lds + buf * (NCH * kBF) + (lane >> 4) * kGS + (wr >> 2) * kPS + (wr >> 3) * 4 + (wr & 3);
#pragma unroll
    for (int j = 0; j < CPW; ++j) {
      const int c = wave + kWB * j;
      f32x4 lv;
      float l4;
      taken(j, lv, l4);
#pragma unroll
      for (int t = 0; t < kK; ++t) {
        float v = tapval(lv, l4, j, t);
        pb[(c >> 4) * kBF + ((c & 15) * kK + t) * 4] = v > 0.f ? v : v * a.xslope;
      }
    }
    }
  };

  const int boff = kgrp * kPS + (kgrp >> 1) * 4 + r16 * 4;
  f32x4 acc[RW][NCH][kNCT], accb[RW];
#pragma unroll
  for (int i = 0; i < RW; ++i) {
    accb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int h = 0; h < NCH; ++h)
#pragma unroll
      for (int j = 0; j < kNCT; ++j) acc[i][h][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const bool with_bias = cch == 0;

  struct Frag {
    f32x4 b[NCH][kNCT];
  };
  int tile = split;
  b_load(tile < a.n_tiles ? tile : (1 << 24));
  b_write(0);
  a_load(tile < a.n_tiles ? tile : (1 << 24));
  __syncthreads();
  // BIAS (chunk group 0's blocks): one more matrix instruction per k-step against a ones operand — two copies of the
  // loop, so that the other blocks carry no branch inside the multiplications
  auto run = [&](auto bias_tag) __attribute__((always_inline)) {
    constexpr bool BIAS = decltype(bias_tag)::value;
    int cur = 0;
    for (; tile < a.n_tiles; tile += a.splits) {
      const int nxt = tile + a.splits < a.n_tiles ? tile + a.splits : (1 << 24);
      // this tile's row fragments (requested a tile ago), then the requests of the next tile: past the last tile every
      // offset is out of range and the loads return zeros nobody uses (no branch: the compiler's wait counts stay exact)
      [[maybe_unused]] f32x4 fa[RW][YB ? 1 : kNG];
      [[maybe_unused]] bf16x8 fab[RW][YB ? NSTEP : 1];
#pragma unroll
      for (int i = 0; i < RW; ++i) {
        if constexpr (YB) {
#pragma unroll
          for (int g = 0; g < NSTEP; ++g) fab[i][g] = a_frag_b(i, g);
        } else {
#pragma unroll
          for (int g = 0; g < kNG; ++g) fa[i][g] = a_frag(i, g);
        }
      }
      a_load(nxt);
      b_load(nxt);
      const float* pb = lds + cur * (NCH * kBF) + boff;
      auto fetch = [&](Frag& f, int g) __attribute__((always_inline)) {
#ifdef RTG_EXP_DW_NOFETCH
        return;
#endif
#pragma unroll
        for (int h = 0; h < NCH; ++h)
#pragma unroll
          for (int j = 0; j < kNCT; ++j)
            f.b[h][j] = *reinterpret_cast<const f32x4*>(pb + h * kBF + g * kGS + j * 64);
      };
      auto mma = [&](const Frag& f, int g) __attribute__((always_inline)) {
#ifdef RTG_EXP_DW_NOMMA
        return;
#endif
        if constexpr (BF) {
#pragma unroll
          for (int i = 0; i < RW; ++i) {
            bf16x8 av;
            if constexpr (YB) {
              av = fab[i][g];
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                av[e] = (__bf16)fa[i][2 * g][e];
                av[4 + e] = (__bf16)fa[i][2 * g + 1][e];
              }
            }
#pragma unroll
            for (int h = 0; h < NCH; ++h)
#pragma unroll
              for (int j = 0; j < kNCT; ++j)
                acc[i][h][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, __builtin_bit_cast(bf16x8, f.b[h][j]), acc[i][h][j], 0, 0, 0);
            if constexpr (BIAS) {
              const __bf16 one = (__bf16)1.0f;
              const bf16x8 ones = {one, one, one, one, one, one, one, one};
              accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, ones, accb[i], 0, 0, 0);
            }
          }
        } else {
#pragma unroll
          for (int kq = 0; kq < 4; ++kq)
#pragma unroll
            for (int i = 0; i < RW; ++i) {
#pragma unroll
              for (int h = 0; h < NCH; ++h)
#pragma unroll
                for (int j = 0; j < kNCT; ++j)
                  acc[i][h][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i][g][kq], f.b[h][j][kq], acc[i][h][j], 0, 0, 0);
              if constexpr (BIAS) accb[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i][g][kq], 1.0f, accb[i], 0, 0, 0);
            }
        }
      };
      // the column fragments of 16-reduction group g + 1 are requested before group g is multiplied (two register sets;
      // one set where twelve fragments are already 48 registers: the 4-chunk shape — and where ten of them sit next to the
      // fp32 row fragments of the bf16-operand instances on fp32 tensors: two sets there are 257 registers, one spilled)
      if constexpr (NCH * kNCT <= 10 && !(BF && !YB && NCH * kNCT == 10)) {
        Frag f0, f1;
        fetch(f0, 0);
#pragma unroll
        for (int g = 0; g < NSTEP; ++g) {
          Frag& fc = (g & 1) ? f1 : f0;
          Frag& fn = (g & 1) ? f0 : f1;
          if (g + 1 < NSTEP) fetch(fn, g + 1);
          __builtin_amdgcn_sched_barrier(0);
          mma(fc, g);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
        Frag f;
#pragma unroll
        for (int g = 0; g < NSTEP; ++g) {
          fetch(f, g);
          mma(f, g);
        }
      }
      if (nxt < a.n_tiles) b_write(cur ^ 1);
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      cur ^= 1;
    }
  };
  if (with_bias) run(std::true_type{});
  else run(std::false_type{});

  // ---- this split's partial: [rows][Cg * K] then the bias partials (gy_scale applied here: the sums are linear in gy)
  float* wpart = a.part + (size_t)split * a.part_stride;
  const int ck = a.Cg * kK;                       // (2-D: Cg = channels x kernel rows)
#ifdef RTG_EXP_DW_NOEPI
  if (acc[0][0][0][0] != 12345.f) return;
#endif
#pragma unroll
  for (int i = 0; i < RW; ++i)
#pragma unroll
    for (int h = 0; h < NCH; ++h)
#pragma unroll
      for (int j = 0; j < kNCT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = m0 + (wave * RW + i) * 16 + kgrp * 4 + r;
          wpart[(size_t)m * ck + (c0 + h * kCch) * kK + j * 16 + r16] = acc[i][h][j][r] * a.gy_scale;
        }
  if (with_bias && r16 == 0) {
    float* bpart = wpart + (size_t)a.Mg * ck;
#pragma unroll
    for (int i = 0; i < RW; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) bpart[m0 + (wave * RW + i) * 16 + kgrp * 4 + r] = accb[i][r] * a.gy_scale;
  }
}

// shape codes 10, 11: (waves, channel chunks per block, 16-row tiles per wave).  Measured and dropped: 4-wave blocks
// (two per CU: 5-15 % slower than the 8-wave block of the same tile) and 32 rows per wave (RW = 2: the two pending
// 16-byte loads per fragment and row tile push the kernel past 256 registers)
struct DwShape {
  int wb, nch, rw;
};
// (the 4-chunk shape: 3-tap layers, or bf16 — half the fragment registers; the 8-chunk shape: 3-tap bf16 layers)
// (a 4-wave block of 64 rows for the 3-tap layers whose row count is no multiple of 128: StftDiscriminator convs.1, 32 -> 64)
constexpr DwShape kDw[] = {{8, 1, 1}, {8, 2, 1}, {8, 4, 1}, {8, 8, 1}, {4, 2, 1}};
constexpr int kNumDw = sizeof(kDw) / sizeof(DwShape);

bool eligible(const RtgWgradDesc* d, int variant) {
  if (variant < 0 || variant >= kNumDw) return false;
  const int kRows = kDw[variant].wb * kDw[variant].rw * 16;
  if (d->groups != 1 || d->C2 != 0 || (d->bf16 != 0 && d->bf16 != 1) || d->dil != 1) return false;
  const bool two_d = d->h_k > 1 || d->h_n > 1;
  if (two_d) {
    // 3 taps at stride 1 / 2, or (round 5: StftDiscriminator along the frequency axis) 5 taps at stride 3
    if (!((d->K == 3 && (d->stride == 1 || d->stride == 2)) || (d->K == 5 && d->stride == 3))) return false;
    if (d->h_in < 1 || d->h_k < 1 || d->h_stride < 1 || d->h_pad < 0 || d->h_n < 1 || d->C1 % d->h_k != 0 || d->B % d->h_n != 0)
      return false;
    if ((long long)(d->B / d->h_n) * (d->C1 / d->h_k) * d->h_in * d->L_in * 4 >= (1ll << 31)) return false;
  } else {
    if (d->K != 5 || (d->stride != 1 && d->stride != 3)) return false;
    if ((long long)d->B * d->C1 * d->L_in * 4 >= (1ll << 31)) return false;
  }
  if (d->Cg != d->C1 || d->Cg % (kCch * kDw[variant].nch) != 0 || d->Mg % kRows != 0) return false;
  if (kDw[variant].nch == 4 && d->K != 3 && !d->bf16) return false;
  if (kDw[variant].nch == 8 && (d->K != 3 || !d->bf16)) return false;
  if (kDw[variant].wb == 4 && (d->K != 3 || d->Mg % 128 == 0)) return false;
  if (d->gy_mode != RTG_PRE_NONE || (d->pre_mode != RTG_PRE_NONE && d->pre_mode != RTG_PRE_LRELU)) return false;
  if (d->Q < 4 || d->Q > d->dy_L) return false;                               // (four consecutive reductions span <= 2 clips)
  // bf16 tensors (RtgWgradDesc.io_bf16): with bf16 operands; a bf16 dy fragment is 8 consecutive reductions
  if (d->io_bf16 < 0 || d->io_bf16 > 3 || (d->io_bf16 != 0 && !d->bf16)) return false;
  if ((d->io_bf16 & 2) && d->Q < 8) return false;
  if ((d->io_bf16 & 1) && d->pre_mode != RTG_PRE_NONE && d->pre_mode != RTG_PRE_LRELU) return false;
  const long long n = (long long)d->B * d->Q;
  if (n >= (1ll << 23)) return false;                                         // float-reciprocal division of n by Q
  if ((long long)d->B * d->Mg * d->dy_L * 4 >= (1ll << 31)) return false;
  return true;
}

}  // namespace

int rtg_dwgrad_variants(void) { return kNumDw; }
int rtg_dwgrad_ok(const RtgWgradDesc* d, int variant) { return eligible(d, variant) ? 1 : 0; }

// split count: a launch lasts (rounds of the chip) x (tiles per block) tile times plus the write and the fixed-order
// read-back of one partial per split
int rtg_dwgrad_splits(const RtgWgradDesc* d, int variant) {
  if (!eligible(d, variant)) return RTG_EINVAL;
  const DwShape sh = kDw[variant];
  const long long base = (long long)(d->Mg / (sh.wb * sh.rw * 16)) * (d->Cg / (kCch * sh.nch));
  const long long tiles = ((long long)d->B * d->Q + kTT - 1) / kTT;
  // (bf16: the matrix part of a tile is an eighth; the operand loads and the staging stay)
  const double t_tile = (d->bf16 ? 0.1 : 0.27) * sh.wb * sh.nch * sh.rw * (d->K / 5.0) + 0.3, t_fixed = 6.0;
  const double t_flush = (double)d->Mg * ((double)d->Cg * d->K + 1) * 8.0 / 3.0e6;
  const long long slots = sh.wb == 8 ? 256 : 512;
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

template <int S, int WB, int NCH, int RW, int K, bool TWO_D, bool BF, int IO = 0>
static int dw_launch_io(const WArgs& a, hipStream_t s) {
  auto k = dwgrad_kernel<S, WB, NCH, RW, K, TWO_D, BF, IO>;
  constexpr int kPS = (kCch * K * 4 + 63) / 64 * 64, kGS = 4 * kPS + 8;     // (the kernel's plane / group strides)
  const size_t lds_bytes = (size_t)2 * NCH * ((BF ? 2 : kNG) * kGS) * sizeof(float);
  static std::atomic<unsigned> optin{0};              // (> 64 KB of dynamic LDS: opt-in per kernel and device)
  if (lds_bytes > 64 * 1024 && rtg_lds_optin((const void*)k, optin) != RTG_OK) return RTG_ERANGE;
  RTG_KLAUNCH(k, dim3((unsigned)(8 * a.per_xcd)), dim3(WB * 64), lds_bytes, s, a);
  return rtg_launch_status();
}
// bf16 operands: on fp32 tensors, or with bf16 dy (io 2: the first dense layer of a stack, whose input is fp32) or bf16 x and
// dy (io 3).  (x bf16 with fp32 dy does not occur: a layer's output gradient is bf16 whenever its input is.)
template <int S, int WB, int NCH, int RW, int K, bool TWO_D, bool BF>
static int dw_launch_bf(const WArgs& a, hipStream_t s) {
  if constexpr (BF) {
    if (a.io == 2) return dw_launch_io<S, WB, NCH, RW, K, TWO_D, true, 2>(a, s);
    if (a.io == 3) return dw_launch_io<S, WB, NCH, RW, K, TWO_D, true, 3>(a, s);
    if (a.io != 0) return RTG_EINVAL;
  }
  return dw_launch_io<S, WB, NCH, RW, K, TWO_D, BF, 0>(a, s);
}
template <int S, int WB, int NCH, int RW, int K, bool TWO_D>
static int dw_launch(const WArgs& a, hipStream_t s) {
  return a.bf ? dw_launch_bf<S, WB, NCH, RW, K, TWO_D, true>(a, s) : dw_launch_bf<S, WB, NCH, RW, K, TWO_D, false>(a, s);
}

int rtg_dwgrad_launch(const RtgWgradDesc* d, int variant, const float* x, const float* dy, float* part, hipStream_t s) {
  if (!eligible(d, variant)) return RTG_EINVAL;
  if (!x || !dy || !part) return RTG_ENULL;
  if ((reinterpret_cast<uintptr_t>(dy) & ((d->io_bf16 & 2) ? 1 : 3)) != 0) return RTG_EINVAL;
  const DwShape sh = kDw[variant];
  WArgs a;
  a.x = x; a.dy = dy; a.part = part; a.bf = d->bf16; a.io = d->io_bf16;
  a.B = d->B; a.Cg = d->Cg; a.L_in = d->L_in; a.Mg = d->Mg; a.Q = d->Q; a.dy_L = d->dy_L; a.pad = d->pad;
  a.xslope = d->pre_mode == RTG_PRE_LRELU ? d->pre_slope : 1.f;
  a.gy_scale = d->gy_scale;
  a.inv_Q = 1.0f / (float)d->Q;
  a.splits = d->splits; a.part_stride = d->part_stride;
  a.n_red = d->B * d->Q;
  a.n_tiles = rtg_ceil_div(a.n_red, kTT);
  a.n_mb = d->Mg / (sh.wb * sh.rw * 16); a.n_cch = d->Cg / (kCch * sh.nch);
  a.per_split = a.n_mb * a.n_cch;
  const long long n_items = (long long)a.per_split * d->splits;
  if (n_items > (1ll << 28)) return RTG_ERANGE;
  a.n_items = (int)n_items;
  a.per_xcd = (int)((n_items + 7) / 8);
  const bool two_d = d->h_k > 1 || d->h_n > 1;
  a.h_in = two_d ? d->h_in : 1; a.h_k = two_d ? d->h_k : 1; a.h_stride = two_d ? d->h_stride : 1;
  a.h_pad = two_d ? d->h_pad : 0; a.h_n = two_d ? d->h_n : 1;
  // (bf16 tensors: + the 16 readable bytes the caller guarantees behind them, see RtgConv1dDesc.io_bf16)
  a.x_bytes = (d->B / a.h_n) * (d->C1 / a.h_k) * a.h_in * d->L_in * ((d->io_bf16 & 1) ? 2 : 4) + ((d->io_bf16 & 1) ? 16 : 0);
  a.dy_bytes = d->B * d->Mg * d->dy_L * ((d->io_bf16 & 2) ? 2 : 4) + ((d->io_bf16 & 2) ? 16 : 0);  // (B = items * h_n)
  const int S = d->stride;
  if (two_d && d->K == 5) {
    if (variant == 0) return dw_launch<3, 8, 1, 1, 5, true>(a, s);
    if (variant == 1) return dw_launch<3, 8, 2, 1, 5, true>(a, s);
    if (variant == 2 && a.bf) return dw_launch_bf<3, 8, 4, 1, 5, true, true>(a, s);
    return RTG_EINVAL;
  }
  if (two_d) {
    if (variant == 0) return S == 1 ? dw_launch<1, 8, 1, 1, 3, true>(a, s) : dw_launch<2, 8, 1, 1, 3, true>(a, s);
    if (variant == 1) return S == 1 ? dw_launch<1, 8, 2, 1, 3, true>(a, s) : dw_launch<2, 8, 2, 1, 3, true>(a, s);
    if (variant == 2) return S == 1 ? dw_launch<1, 8, 4, 1, 3, true>(a, s) : dw_launch<2, 8, 4, 1, 3, true>(a, s);
    if (variant == 4) return S == 1 ? dw_launch<1, 4, 2, 1, 3, true>(a, s) : dw_launch<2, 4, 2, 1, 3, true>(a, s);
    return S == 1 ? dw_launch_bf<1, 8, 8, 1, 3, true, true>(a, s) : dw_launch_bf<2, 8, 8, 1, 3, true, true>(a, s);
  }
  if (variant == 0) return S == 1 ? dw_launch<1, 8, 1, 1, 5, false>(a, s) : dw_launch<3, 8, 1, 1, 5, false>(a, s);
  if (variant == 1) return S == 1 ? dw_launch<1, 8, 2, 1, 5, false>(a, s) : dw_launch<3, 8, 2, 1, 5, false>(a, s);
  if (variant == 2 && a.bf) return S == 1 ? dw_launch_bf<1, 8, 4, 1, 5, false, true>(a, s) : dw_launch_bf<3, 8, 4, 1, 5, false, true>(a, s);
  return RTG_EINVAL;
}
