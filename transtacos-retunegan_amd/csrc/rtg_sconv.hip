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
constexpr int kMaxIt = 6;                              // staged virtual positions per lane (<= 384 per wave and chunk)

struct SArgs {
  const float *x1, *x2, *wp, *bias, *mask, *res;
  float *out, *out2;
  int B, C1, C2, L, Mg, n_cc, n_c1, K, dil, pad, out_C, out_split;
  int pre, act, accumulate;
  float pre_slope, mask_slope, out_scale, act_slope;
  int n_cols, n_mt, SEG, nvp, PS;                      // columns, row tiles, positions per clip segment, staged positions, plane stride
};

__device__ __forceinline__ float sc_load(rsrc_t r, unsigned off, unsigned soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, soff, 0));
}

template <int KS>
__global__ __launch_bounds__(KS * 64) void sconv_kernel(const SArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];      // [KS][4 planes][PS] | [KS][16][64] partial tiles
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n16 = lane & 15, kgrp = lane >> 4;
  const int mt = (int)blockIdx.x % a.n_mt, ct = (int)blockIdx.x / a.n_mt;     // the row tiles of one column tile are neighbours
  const int n0 = ct * 64;
  const int cA = n0 / a.L;                              // first clip of the block's columns
  const int n_last = (n0 + 63 < a.n_cols ? n0 + 63 : a.n_cols - 1);
  const int nvp = (n_last / a.L - cA + 1) * a.SEG;     // staged positions of THIS block: whole segments of the clips it touches
  float* pl = lds + wave * (4 * a.PS);
  float* red = lds + KS * (4 * a.PS);

  // ---- this lane's four columns: staged position of tap 0
  int vp[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    int n = n0 + 16 * j + n16;
    if (n > a.n_cols - 1) n = a.n_cols - 1;            // junk column: a valid position, dropped in the epilogue
    const int clip = n / a.L, q = n - clip * a.L;
    vp[j] = (clip - cA) * a.SEG + q;
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

  f32x4 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
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
      f32x4 bf[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) bf[j] = *reinterpret_cast<const f32x4*>(pl + kgrp * a.PS + (vp[j] + t * a.dil) * 4);
#pragma unroll
      for (int kq = 0; kq < 4; ++kq)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[kq], bf[j][kq], acc[j], 0, 0, 0);
    }
  }
  // ---- the KS partial tiles meet in LDS; wave 0 adds them in fixed order and runs the epilogue
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[(wave * 16 + j * 4 + r) * 64 + lane] = acc[j][r];
  __syncthreads();
  if (wave != 0) return;
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float s_ = red[(j * 4 + r) * 64 + lane];
      for (int w = 1; w < KS; ++w) s_ += red[(w * 16 + j * 4 + r) * 64 + lane];
      acc[j][r] = s_;
    }
  // out = act(((acc + bias) * dmask + res) * out_scale) (+ out): the epilogue arithmetic of the general kernel
  const float mslope = a.mask ? a.mask_slope : 1.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = n0 + 16 * j + n16;
    if (n >= a.n_cols) continue;
    const int clip = n / a.L, q = n - clip * a.L;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = mt * 16 + 4 * kgrp + r;
      if (m >= a.Mg) continue;
      float* dst = a.out;
      int chd = m, Cd = a.out_C;
      if (a.out_split) {
        if (m >= a.out_split) { dst = a.out2; chd = m - a.out_split; Cd = a.out_C - a.out_split; }
        else Cd = a.out_split;
      }
      if (!dst) continue;
      const size_t o = ((size_t)clip * Cd + chd) * a.L + q;
      float v = acc[j][r] + (a.bias ? a.bias[m] : 0.f);
      const float mv = a.mask ? a.mask[o] : 1.f, rv = a.res ? a.res[o] : 0.f;
      v = __builtin_fmaf(v, mv > 0.f ? 1.f : mslope, rv) * a.out_scale;
      if (a.act == RTG_ACT_LRELU) v = rtg_lrelu(v, a.act_slope);
      else if (a.act == RTG_ACT_TANH) v = tanhf(v);
      if (a.accumulate) v += dst[o];
      dst[o] = v;
    }
  }
}

bool sconv_eligible(const RtgConv1dDesc* d) {
  if (!d->wp16 || d->groups != 1 || d->tap_major || d->bf16 || d->stride != 1 || d->shuf_S != 1) return false;
  if (d->h_k > 1 || d->h_n > 1 || d->dil < 1 || d->K < 1 || d->K > 8) return false;
  if (d->pre_mode != RTG_PRE_NONE && d->pre_mode != RTG_PRE_LRELU) return false;
  if (d->C1 % 16 != 0 || d->C2 % 16 != 0 || d->C1 + d->C2 != d->Cg || d->Mg % 16 != 0 || d->Mg != d->out_C) return false;
  if (d->out_split % 16 != 0 || d->out_split < 0 || d->out_split >= d->out_C) return false;
  if (d->L_in < 32 || d->L_in > 64 || d->Q != d->L_in || d->out_L != d->L_in) return false;
  if (d->pad < 0 || d->pad > d->dil * (d->K - 1)) return false;      // (left padding; the right one follows from Q = L_in)
  const long long n_cols = (long long)d->B * d->L_in;
  if (n_cols > 8192 || d->Cg < 64) return false;       // (more columns fill the chip in the general kernel's tiling)
  const int seg = d->L_in + (d->K - 1) * d->dil;
  if (((d->L_in + 62) / d->L_in + 1) * seg > 64 * kMaxIt) return false;
  if ((long long)d->B * d->Cg * d->L_in * 4 >= (1ll << 31) || (long long)d->B * d->out_C * d->L_in * 4 >= (1ll << 31)) return false;
  return true;
}

}  // namespace

#define RTG_SCONV_CODE 9000

int rtg_sconv_candidates(const RtgConv1dDesc* d, int* codes, int max) {
  if (RTG_ENV_INT("RTG_SCONV", 1) == 0 || !sconv_eligible(d)) return 0;
  int cnt = 0;
  const int n_cc = d->Cg / 16;
  if (n_cc >= 8 && cnt < max) codes[cnt++] = RTG_SCONV_CODE + 8;
  if (n_cc >= 4 && cnt < max) codes[cnt++] = RTG_SCONV_CODE + 4;
  return cnt;
}

int rtg_sconv_launch(const RtgConv1dDesc* d, int code, const float* x1, const float* x2, const float* wp, const float* bias,
                     const float* mask, const float* res, float* out, float* out2, hipStream_t s) {
  if (!sconv_eligible(d)) return RTG_EINVAL;
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
  a.B = d->B; a.C1 = d->C1; a.C2 = d->C2; a.L = d->L_in; a.Mg = d->Mg; a.n_cc = d->Cg / 16; a.n_c1 = d->C1 / 16;
  a.K = d->K; a.dil = d->dil; a.pad = d->pad; a.out_C = d->out_C; a.out_split = d->out_split;
  a.pre = d->pre_mode == RTG_PRE_LRELU ? 1 : 0; a.act = d->act; a.accumulate = d->accumulate;
  a.pre_slope = d->pre_slope; a.mask_slope = d->mask_slope; a.out_scale = d->out_scale; a.act_slope = d->act_slope;
  a.n_cols = d->B * d->L_in;
  a.n_mt = d->Mg / 16;
  a.SEG = d->L_in + (d->K - 1) * d->dil;
  a.nvp = ((d->L_in + 62) / d->L_in + 1) * a.SEG;      // the most clips a 64-column tile can touch, whole segments
  a.PS = ((a.nvp * 4 + 63) / 64) * 64;                 // planes a multiple of 256 bytes apart (rtg_dconv.hip: LDS banking)
  const size_t lds_bytes = ((size_t)ks * 4 * a.PS + (size_t)ks * 16 * 64) * sizeof(float);
  if (lds_bytes > 150 * 1024) return RTG_ERANGE;
  const unsigned blocks = (unsigned)(a.n_mt * rtg_ceil_div(a.n_cols, 64));
  if (ks == 4) {
    static bool attr4 = false;
    if (lds_bytes > 64 * 1024 && !attr4) {
      if (hipFuncSetAttribute((const void*)sconv_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return RTG_ERANGE;
      attr4 = true;
    }
    RTG_KLAUNCH((sconv_kernel<4>), dim3(blocks), dim3(256), lds_bytes, s, a);
  } else {
    static bool attr8 = false;
    if (lds_bytes > 64 * 1024 && !attr8) {
      if (hipFuncSetAttribute((const void*)sconv_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return RTG_ERANGE;
      attr8 = true;
    }
    RTG_KLAUNCH((sconv_kernel<8>), dim3(blocks), dim3(512), lds_bytes, s, a);
  }
  return rtg_launch_status();
}
