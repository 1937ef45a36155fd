// rtg_sconv.hip — stride-1 "same" convolutions over FEW columns with split-K over the waves of a block (round 4): the
// bottom of the UNet generator (retunegan/models/generator.py:734-788 at T / 256 = 32 positions per clip): conv_fuse
// (208 -> 256, k7, the concatenated [mel | encoder] input, generator.py:755) forward and backward-data, the six dilated k3
// convs of the 128-channel ResidualStack (generator.py:33-77).
//
// At batch 32 these layers have 1024 output columns.  The general kernel tiles them 64 x 64 per block of four waves: 64-128
// blocks on 256 CUs, and every wave walks the WHOLE reduction (13 chunks x 7 taps: a serial chain of ~1450 matrix
// instructions, 19 us at full rate, one wave per SIMD with nothing to overlap) — 64-80 us per launch at 10-12 TFLOP/s, on
// the critical path of the step (the generator is a serial chain).  Here a block is ONE 16-row tile x 64 columns and its
// KS waves split the 16-channel chunks of the reduction (chunk cc goes to wave cc % KS); the KS partial tiles meet in LDS
// and are added in fixed order (wave 0, 1, ...): 256-1024 blocks, chains of ~100-360 instructions.  Operands as in
// rtg_dconv.hip: the 16-byte-fragment weight image (RtgConv1dDesc.wp16; one coalesced 1-KB load per chunk and tap) and the
// staged patch as four planes [kgrp][position][kq] read with ds_read_b128; every wave stages its own chunks (its own LDS
// planes: no barrier inside the reduction).  The summation order differs from the general kernel's (rounding level):
// block-shape codes 9000 + KS, listed by rtg_conv1d_tile_candidates for eligible problems, timed by the tuner.
#include "rtg_common.h"

namespace {

using rsrc_t = __amdgpu_buffer_rsrc_t;
#define SC_OOB 0x80000000u

struct SArgs {
  const float *x1, *x2, *wp, *bias, *mask, *res;
  float *out, *out2;
  int B, C1, C2, L, Q, S, out_L, shuf_S, shuf_P, Mg, n_cc, n_c1, K, dil, pad, out_C, out_split;   // L: input row length
  int pre, act, accumulate;
  float pre_slope, mask_slope, out_scale, act_slope;
  int n_cols, n_mt, SEG, nvp, PS;                      // columns, row tiles, positions per clip segment, staged positions, plane stride
};

__device__ __forceinline__ float sc_load(rsrc_t r, unsigned off, unsigned soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, soff, 0));
}

// KS: waves (slices of the reduction); NT: 16-column tiles per block; MAXIT: staged positions per lane and chunk / 64
template <int KS, int NT, int MAXIT>
__global__ __launch_bounds__(KS * 64) void sconv_kernel(const SArgs a) {
  constexpr int kMaxIt = MAXIT, BN = NT * 16;
  extern __shared__ __attribute__((aligned(16))) float lds[];      // [KS][4 planes][PS] | [KS][16][64] partial tiles
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n16 = lane & 15, kgrp = lane >> 4;
  const int mt = (int)blockIdx.x % a.n_mt, ct = (int)blockIdx.x / a.n_mt;     // the row tiles of one column tile are neighbours
  const int n0 = ct * BN;
  const int cA = n0 / a.Q;                              // first clip of the block's columns
  const int n_last = (n0 + BN - 1 < a.n_cols ? n0 + BN - 1 : a.n_cols - 1);
  const int nvp = (n_last / a.Q - cA + 1) * a.SEG;     // staged positions of THIS block: whole segments of the clips it touches
  float* pl = lds + wave * (4 * a.PS);
  float* red = lds + KS * (4 * a.PS);

  // ---- this lane's NT columns: staged position of tap 0 (column q of a clip reads input positions q * S + t * dil - pad)
  int vp[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    int n = n0 + 16 * j + n16;
    if (n > a.n_cols - 1) n = a.n_cols - 1;            // junk column: a valid position, dropped in the epilogue
    const int clip = n / a.Q, q = n - clip * a.Q;
    vp[j] = (clip - cA) * a.SEG + q * a.S;
  }
  // ---- staging geometry: virtual position v = lane + 64 it <-> (clip, input position)
  unsigned s1[kMaxIt], s2[kMaxIt];                      // byte offsets of (clip, channel 0, position) in x1 / x2, or out of range
#pragma unroll
  for (int it = 0; it < kMaxIt; ++it) {
    const int v = lane + 64 * it;
    const int seg = v / a.SEG, w = v - seg * a.SEG;
    const int clip = cA + seg, pos = w - a.pad;
    const bool ok = v < nvp && clip < a.B && pos >= 0 && pos < a.L;
    s1[it] = ok ? (unsigned)(clip * a.C1 * a.L + pos) * 4u : SC_OOB;
    s2[it] = ok ? (unsigned)(clip * a.C2 * a.L + pos) * 4u : SC_OOB;
  }
  const rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc((void*)a.x1, 0, a.B * a.C1 * a.L * 4, 0x00020000);
  const rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x2 ? a.x2 : a.x1), 0, a.x2 ? a.B * a.C2 * a.L * 4 : 0, 0x00020000);
  const unsigned chb = (unsigned)a.L * 4u;
  const float slope = a.pre ? a.pre_slope : 1.f;

  f32x4 acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const f32x4* wbase = reinterpret_cast<const f32x4*>(a.wp) + (size_t)mt * a.n_cc * a.K * 64 + lane;

  for (int cc = wave; cc < a.n_cc; cc += KS) {
    // ---- stage the chunk's 16 channels: plane g holds channels g, 4 + g, 8 + g, 12 + g of every position (16 bytes)
    const bool from2 = cc >= a.n_c1;
    const rsrc_t rx = from2 ? r2 : r1;
    const unsigned cbase = (unsigned)((from2 ? cc - a.n_c1 : cc) * 16) * chb;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float st[4][kMaxIt];
#pragma unroll
      for (int kq = 0; kq < 4; ++kq)
#pragma unroll
        for (int it = 0; it < kMaxIt; ++it) {
          const unsigned so = from2 ? s2[it] : s1[it];
          st[kq][it] = (lane + 64 * it < nvp) ? sc_load(rx, so, cbase + (unsigned)(4 * kq + g) * chb) : 0.f;
        }
#pragma unroll
      for (int it = 0; it < kMaxIt; ++it) {
        const int v = lane + 64 * it;
        if (v < nvp) {
          f32x4 w4;
#pragma unroll
          for (int kq = 0; kq < 4; ++kq) {
            const float t = st[kq][it];
            w4[kq] = t > 0.f ? t : t * slope;
          }
          *reinterpret_cast<f32x4*>(pl + g * a.PS + v * 4) = w4;
        }
      }
    }
    // ---- K taps x 4 column tiles x 4 k-steps (the wave's own LDS writes above are in order with these reads)
    const f32x4* wp = wbase + (size_t)cc * a.K * 64;
    for (int t = 0; t < a.K; ++t) {
      const f32x4 af = wp[(size_t)t * 64];
      f32x4 bf[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j) bf[j] = *reinterpret_cast<const f32x4*>(pl + kgrp * a.PS + (vp[j] + t * a.dil) * 4);
#pragma unroll
      for (int kq = 0; kq < 4; ++kq)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[kq], bf[j][kq], acc[j], 0, 0, 0);
    }
  }
  // ---- the KS partial tiles meet in LDS; wave 0 adds them in fixed order and runs the epilogue
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[(wave * NT * 4 + j * 4 + r) * 64 + lane] = acc[j][r];
  __syncthreads();
  if (wave != 0) return;
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float s_ = red[(j * 4 + r) * 64 + lane];
      for (int w = 1; w < KS; ++w) s_ += red[(w * NT * 4 + j * 4 + r) * 64 + lane];
      acc[j][r] = s_;
    }
  // out = act(((acc + bias) * dmask + res) * out_scale) (+ out): the epilogue arithmetic of the general kernel
  const float mslope = a.mask ? a.mask_slope : 1.f;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = n0 + 16 * j + n16;
    if (n >= a.n_cols) continue;
    const int clip = n / a.Q, q = n - clip * a.Q;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = mt * 16 + 4 * kgrp + r;
      if (m >= a.Mg) continue;
      float* dst = a.out;
      // GEMM row m is output channel m / shuf_S at phase m % shuf_S (polyphase backward-data / transposed conv): position
      // q * shuf_S + phase - shuf_P of the output row; shuf_S == 1: channel m, position q
      int chd = m, Cd = a.out_C, pos = q;
      if (a.shuf_S > 1) {
        chd = m / a.shuf_S;
        pos = q * a.shuf_S + (m - chd * a.shuf_S) - a.shuf_P;
      }
      if (pos < 0 || pos >= a.out_L) continue;
      const float bv = a.bias ? a.bias[chd] : 0.f;        // (indexed by the output channel, include/rtg.h: shuffle store)
      if (a.out_split) {
        if (chd >= a.out_split) { dst = a.out2; chd -= a.out_split; Cd = a.out_C - a.out_split; }
        else Cd = a.out_split;
      }
      if (!dst) continue;
      const size_t o = ((size_t)clip * Cd + chd) * a.out_L + pos;
      float v = acc[j][r] + bv;
      const float mv = a.mask ? a.mask[o] : 1.f, rv = a.res ? a.res[o] : 0.f;
      v = __builtin_fmaf(v, mv > 0.f ? 1.f : mslope, rv) * a.out_scale;
      if (a.act == RTG_ACT_LRELU) v = rtg_lrelu(v, a.act_slope);
      else if (a.act == RTG_ACT_TANH) v = tanhf(v);
      if (a.accumulate) v += dst[o];
      dst[o] = v;
    }
  }
}

// the instance (NT, MAXIT) that serves the descriptor: 1 = <4 tiles, 6>, 2 = <2 tiles, 9>; 0 = none
int sconv_kind(const RtgConv1dDesc* d) {
  // (wp16 == 2: the layer's owner allows this kernel's summation order next to the bit-identical block shapes, rtg/bank.py)
  if (d->wp16 != 2 || d->groups != 1 || d->tap_major || d->bf16 || d->stride < 1 || d->stride > 8) return 0;
  if (d->h_k > 1 || d->h_n > 1 || d->dil < 1 || d->K < 1 || d->K > 16 || d->shuf_S < 1 || d->shuf_S > 8) return 0;
  if (d->pre_mode != RTG_PRE_NONE && d->pre_mode != RTG_PRE_LRELU) return 0;
  if (d->C1 % 16 != 0 || d->C2 % 16 != 0 || d->C1 + d->C2 != d->Cg || d->Mg % 16 != 0 || d->Mg != d->out_C * d->shuf_S) return 0;
  if (d->out_split % 16 != 0 || d->out_split < 0 || d->out_split >= d->out_C) return 0;
  if (d->Q < 16 || d->Q > 64 || d->L_in < 1 || d->out_L < 1) return 0;
  if (d->pad < 0) return 0;
  const long long n_cols = (long long)d->B * d->Q;
  if (n_cols > 8192 || d->Cg < 64) return 0;           // (more columns fill the chip in the general kernel's tiling)
  const long long seg = (long long)(d->Q - 1) * d->stride + (long long)(d->K - 1) * d->dil + 1;
  if ((long long)d->B * d->Cg * d->L_in * 4 >= (1ll << 31) || (long long)d->B * d->out_C * d->out_L * 4 >= (1ll << 31)) return 0;
  // the clips a block's columns can touch, whole segments each
  if (((d->Q + 62) / d->Q + 1) * seg <= 64 * 6) return 1;
  if (((d->Q + 30) / d->Q + 1) * seg <= 64 * 9) return 2;
  return 0;
}

// dynamic LDS of a block of ks waves: every wave's four patch planes, then the ks partial tiles
struct SGeo {
  int bn, nt, seg, nvp, ps;
  size_t lds_bytes;
};
SGeo sconv_geo(const RtgConv1dDesc* d, int kind, int ks) {
  SGeo g;
  g.bn = kind == 1 ? 64 : 32;
  g.nt = g.bn / 16;
  g.seg = (d->Q - 1) * d->stride + (d->K - 1) * d->dil + 1;
  g.nvp = ((d->Q + g.bn - 2) / d->Q + 1) * g.seg;      // the most clips a block's columns can touch, whole segments
  g.ps = ((g.nvp * 4 + 63) / 64) * 64;                 // planes a multiple of 256 bytes apart (rtg_dconv.hip: LDS banking)
  g.lds_bytes = ((size_t)ks * 4 * g.ps + (size_t)ks * g.nt * 4 * 64) * sizeof(float);
  return g;
}
constexpr size_t kSconvLdsMax = 150 * 1024;

}  // namespace

#define RTG_SCONV_CODE 9000

int rtg_sconv_candidates(const RtgConv1dDesc* d, int* codes, int max) {
  const int kind = sconv_kind(d);
  if (!kind) return 0;
  int cnt = 0;
  const int n_cc = d->Cg / 16;
  if (n_cc >= 8 && cnt < max && sconv_geo(d, kind, 8).lds_bytes <= kSconvLdsMax) codes[cnt++] = RTG_SCONV_CODE + 8;
  if (n_cc >= 4 && cnt < max && sconv_geo(d, kind, 4).lds_bytes <= kSconvLdsMax) codes[cnt++] = RTG_SCONV_CODE + 4;
  return cnt;
}

template <int KS, int NT, int MAXIT>
static int sconv_go(const SArgs& a, unsigned blocks, size_t lds_bytes, hipStream_t s) {
  auto k = sconv_kernel<KS, NT, MAXIT>;
  static std::atomic<unsigned> optin{0};              // (> 64 KB of dynamic LDS: opt-in per kernel and device)
  if (lds_bytes > 64 * 1024 && rtg_lds_optin((const void*)k, optin) != RTG_OK) return RTG_ERANGE;
  RTG_KLAUNCH(k, dim3(blocks), dim3(KS * 64), lds_bytes, s, a);
  return rtg_launch_status();
}

int rtg_sconv_launch(const RtgConv1dDesc* d, int code, const float* x1, const float* x2, const float* wp, const float* bias,
                     const float* mask, const float* res, float* out, float* out2, hipStream_t s) {
  const int kind = sconv_kind(d);
  if (!kind) return RTG_EINVAL;
  const int ks = code - RTG_SCONV_CODE;
  if (ks != 4 && ks != 8) return RTG_EINVAL;
  if (!x1 || !wp || (!out && !out2) || (d->C2 > 0 && !x2)) return RTG_ENULL;
  if (d->out_split && (mask || res || d->accumulate)) return RTG_EINVAL;     // (shaped like ONE output tensor)
  if (!d->out_split && !out) return RTG_ENULL;
  if ((reinterpret_cast<uintptr_t>(wp) & 15) != 0) return RTG_EINVAL;
  SArgs a;
  // the 16-byte-fragment image follows the standard image of the layer (RtgConv1dDesc.wp16)
  const long long std_size = rtg_packed_size(1, d->Mg, d->Cg, d->K, d->tile_m);
  if (std_size < 0 || (std_size & 3) != 0) return RTG_EINVAL;
  a.x1 = x1; a.x2 = x2; a.wp = wp + std_size; a.bias = bias; a.mask = mask; a.res = res; a.out = out; a.out2 = out2;
  a.B = d->B; a.C1 = d->C1; a.C2 = d->C2; a.L = d->L_in; a.Q = d->Q; a.S = d->stride; a.out_L = d->out_L;
  a.shuf_S = d->shuf_S; a.shuf_P = d->shuf_P;
  a.Mg = d->Mg; a.n_cc = d->Cg / 16; a.n_c1 = d->C1 / 16;
  a.K = d->K; a.dil = d->dil; a.pad = d->pad; a.out_C = d->out_C; a.out_split = d->out_split;
  a.pre = d->pre_mode == RTG_PRE_LRELU ? 1 : 0; a.act = d->act; a.accumulate = d->accumulate;
  a.pre_slope = d->pre_slope; a.mask_slope = d->mask_slope; a.out_scale = d->out_scale; a.act_slope = d->act_slope;
  a.n_cols = d->B * d->Q;
  a.n_mt = d->Mg / 16;
  const SGeo g = sconv_geo(d, kind, ks);
  a.SEG = g.seg; a.nvp = g.nvp; a.PS = g.ps;
  const size_t lds_bytes = g.lds_bytes;
  if (lds_bytes > kSconvLdsMax) return RTG_ERANGE;
  const unsigned blocks = (unsigned)(a.n_mt * rtg_ceil_div(a.n_cols, g.bn));
  if (kind == 1) return ks == 4 ? sconv_go<4, 4, 6>(a, blocks, lds_bytes, s) : sconv_go<8, 4, 6>(a, blocks, lds_bytes, s);
  return ks == 4 ? sconv_go<4, 2, 9>(a, blocks, lds_bytes, s) : sconv_go<8, 2, 9>(a, blocks, lds_bytes, s);
}
