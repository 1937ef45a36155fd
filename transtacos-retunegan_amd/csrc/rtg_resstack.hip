// rtg_resstack.hip — a whole ResidualStack (generator.py:33-77) of the UNet-G encoder in ONE launch per direction.
//
//   forward   for d in (1, 3, 9):  r = conv_d(lrelu(x));  x = x + conv_1(lrelu(r))          [+ lrelu on the last x]
//   backward  g = dy * act'(y);  for d in (9, 3, 1):  g_r = lrelu'(r) * conv_1^T(g);  g = g + lrelu'(x_prev) * conv_d^T(g_r)
// Six k=3 convolutions of C x C channels over clips of 32 (C = 128) or 256 (C = 64) samples: 5 % of the generator's MACs,
// but 36 launches of the general kernel (6 forward, 6 backward-data per stack) that each sit on its fixed cost of
// 15-30 us for 0.1-0.2 GFLOP.  A clip fits in LDS: a block keeps the running tensor of ONE clip in two LDS buffers and walks
// the six layers with a barrier in between; every layer's result also goes to HBM (the forward's intermediates are the
// backward's masks and the weight gradients' inputs; the backward's are the weight gradients' output cotangents), so
// the weight-gradient launches stay as they are.
//   block   8 waves.  C = 128: wave = (row tile of 32 channels, half of the 8 channel chunks), the two halves meet in LDS;
//           C = 64: wave = (row tile, two column tiles of 32 positions).  v_mfma_f32_32x32x2_f32, 96 A fragments per wave
//           and layer, held in registers and requested one layer ahead (straight from the packed weights of rtg_conv1d).
//   layers  even: buffer A -> buffer B; odd: B -> A with the residual (A itself) added in place.  The same kernel runs the
//           backward: backward-data = the same convolution on the flipped (RTG_PACK_DGRAD_S1) weights, the leaky-relu
//           derivative masks come from the saved forward tensors.
#include <stdlib.h>

#include <type_traits>

#include "rtg_common.h"

namespace {

struct StackArgs {
  const float* in;              // x (forward) or dy (backward)
  const float* pro_aux;         // backward: y, for the derivative of the output activation (or null)
  const float* wp[6];           // packed weights per layer, in execution order
  const float* bias[6];         // forward only
  const float* mask[6];         // backward only: the forward tensor whose sign masks this layer's result
  float* gout[6];               // every layer's result in HBM
  int dil[6];
  int B;
  float pre_slope;              // leaky-relu of every conv input (forward), 1 in the backward
  float mask_slope;
  int final_act;                // forward: leaky-relu on the last result; backward: its derivative in the prologue
  float act_slope;
  int L, n_t;                   // clip length, position tiles per clip
};

__device__ __forceinline__ int mrow32(int lane, int r) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

constexpr int kStackWaves = 8;
constexpr int kHalo = 9;

// C channels; a block computes a REGION of R positions of one clip and stores the CEN central ones (R - CEN = 32: the
// stack's receptive field is 16 positions to each side; R == CEN: the whole clip, no neighbours).  Every layer is computed on
// the whole region: what the halo misses only reaches 16 positions inward, never the centre.
template <int C, int R, int CEN>
__global__ __launch_bounds__(64 * kStackWaves) void resstack_kernel(const StackArgs a) {
  constexpr int NCC = C / RTG_CK;                 // 16-channel chunks
  constexpr int RT = C / 32;                      // row tiles
  constexpr int CT = R / 32;                      // column tiles of the region
  constexpr int NKS = kStackWaves / (RT * CT);    // K splits across waves (C = 128: 2)
  constexpr int CHP = NCC / NKS;                  // chunks per wave
  constexpr int NA = CHP * 3 * 8;                 // A fragments per wave and layer
  constexpr int LP = R + 2 * kHalo + 2;           // LDS row pitch (zero columns on both sides: the convs' zero padding)
  constexpr int HL = (R - CEN) / 2;               // halo positions on each side of the centre
  static_assert(RT * CT * NKS == kStackWaves && NCC % NKS == 0 && NKS <= 2, "wave decomposition");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* bufA = lds;
  float* bufB = lds + C * LP;
  float* scr = lds + 2 * C * LP;                  // K-split meeting point: [RT][32 x 32]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kk = lane >> 5, n_lane = lane & 31;
  const int rt = wave % RT;
  const int ks = NKS == 2 ? wave / RT : 0;
  const int ct = NKS == 2 ? 0 : wave / RT;
  const int b = blockIdx.x / a.n_t, ti = blockIdx.x - b * a.n_t;
  const int t0 = ti * CEN;                        // first central position
  const int rs = t0 - HL;                         // clip position of region column 0

  // ---- A fragments of a layer: ((chunk, tap), channel pair) -> 64 consecutive floats in the packed layout.  Two register
  // sets where they fit (the next layer's fragments are requested a layer ahead), else requested at the top of the layer
  // and waited for one by one as the MFMA loop reaches them.
  constexpr bool DB = NA <= 96 && C != 64;
  float A0[NA], A1[DB ? NA : 1];
  auto aload = [&](int l, float (&Af)[NA]) __attribute__((always_inline)) {
    const float* w = a.wp[l] + ((size_t)(rt * NCC + ks * CHP) * 3) * 8 * 64 + lane;
#pragma unroll
    for (int f = 0; f < NA; ++f) Af[f] = w[f * 64];
  };
  aload(0, A0);

  // ---- buffers: zero (the edge columns stay zero), then the region of the clip (zeros outside the clip)
  for (int i = tid; i < 2 * C * LP; i += 64 * kStackWaves) lds[i] = 0.f;
  __syncthreads();
  for (int i = tid; i < C * R; i += 64 * kStackWaves) {
    const int c = i / R, p = i - c * R;
    const int t = rs + p;
    if (t < 0 || t >= a.L) continue;
    const size_t gi = ((size_t)b * C + c) * a.L + t;
    float v = a.in[gi];
    if (a.pro_aux && a.final_act) v *= (a.pro_aux[gi] > 0.f ? 1.f : a.act_slope);
    bufA[c * LP + kHalo + p] = v;
  }
  __syncthreads();

  auto layer = [&](auto LI, auto& Acur, auto& Anext) __attribute__((always_inline)) {
    constexpr int l = decltype(LI)::value;
    constexpr bool odd = (l & 1) != 0;
    const float* in = odd ? bufB : bufA;
    float* outb = odd ? bufA : bufB;
    if constexpr (DB) {
      if (l + 1 < 6) aload(l + 1, Anext);          // the next layer's fragments travel during this layer's MFMAs
    } else {
      if (l > 0) aload(l, Acur);
    }
    const int dil = a.dil[l];
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float* bp = in + (ks * CHP * RTG_CK + kk) * LP + kHalo + ct * 32 + n_lane - dil;
    constexpr int NS = NA, PD = 4;                 // B-fragment reads requested PD steps (256 cycles of MFMAs) ahead
    float vb[PD + 1];
    auto bload = [&](int s_) __attribute__((always_inline)) {
      const int cci = s_ / 24, tap = (s_ / 8) % 3, cp = s_ % 8;
      return bp[(cci * RTG_CK + cp * 2) * LP + tap * dil];
    };
#pragma unroll
    for (int s_ = 0; s_ < PD; ++s_) vb[s_] = bload(s_);
#pragma unroll
    for (int s_ = 0; s_ < NS; ++s_) {
      if (s_ + PD < NS) vb[(s_ + PD) % (PD + 1)] = bload(s_ + PD);
      __builtin_amdgcn_sched_barrier(0);
      float v = vb[s_ % (PD + 1)];
      v = v > 0.f ? v : v * a.pre_slope;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Acur[s_], v, acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (NKS == 2) {                                // the upper channel half hands its sums to the lower one
      if (ks == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) scr[rt * 1024 + r * 64 + lane] = acc[r];
      }
      __syncthreads();
      if (ks == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] += scr[rt * 1024 + r * 64 + lane];
      }
    }
    __syncthreads();                               // every wave is done reading `in` (and, odd layers, may overwrite bufA)
    if (ks == 0) {
      const int pos = ct * 32 + n_lane;            // region column
      const int t = rs + pos;                      // clip position
      const bool inside = t >= 0 && t < a.L;       // outside the clip the next layer must see zero padding
      const bool central = inside && t >= t0 && t < t0 + CEN;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = rt * 32 + mrow32(lane, r);
        const size_t gi = ((size_t)b * C + m) * a.L + (inside ? t : 0);
        float v = acc[r] + (a.bias[l] ? a.bias[l][m] : 0.f);
        const float mf = (a.mask[l] && inside) ? (a.mask[l][gi] > 0.f ? 1.f : a.mask_slope) : 1.f;
        const float rv = odd ? outb[m * LP + kHalo + pos] : 0.f;
        v = __builtin_fmaf(v, mf, rv);
        if (l < 5) outb[m * LP + kHalo + pos] = inside ? v : 0.f;
        if (l == 5 && a.final_act && !a.pro_aux) v = v > 0.f ? v : v * a.act_slope;
        if (central) a.gout[l][gi] = v;
      }
    }
    __syncthreads();
  };
  if constexpr (DB) {
    layer(std::integral_constant<int, 0>{}, A0, A1);
    layer(std::integral_constant<int, 1>{}, A1, A0);
    layer(std::integral_constant<int, 2>{}, A0, A1);
    layer(std::integral_constant<int, 3>{}, A1, A0);
    layer(std::integral_constant<int, 4>{}, A0, A1);
    layer(std::integral_constant<int, 5>{}, A1, A0);
  } else {
    layer(std::integral_constant<int, 0>{}, A0, A0);
    layer(std::integral_constant<int, 1>{}, A0, A0);
    layer(std::integral_constant<int, 2>{}, A0, A0);
    layer(std::integral_constant<int, 3>{}, A0, A0);
    layer(std::integral_constant<int, 4>{}, A0, A0);
    layer(std::integral_constant<int, 5>{}, A0, A0);
  }
}

template <int C, int R, int CEN>
int launch(StackArgs a, int L, hipStream_t s) {
  constexpr int LP = R + 2 * kHalo + 2;
  constexpr int NKS = kStackWaves / ((C / 32) * (R / 32));
  const size_t lds_bytes = ((size_t)2 * C * LP + (NKS == 2 ? 4 * 1024 : 0)) * sizeof(float);
  static std::atomic<unsigned> optin{0};              // (> 64 KB of dynamic LDS: opt-in per kernel and device)
  if (rtg_lds_optin(reinterpret_cast<const void*>(&resstack_kernel<C, R, CEN>), optin) != RTG_OK) return RTG_ERANGE;
  a.L = L;
  a.n_t = (L + CEN - 1) / CEN;
  RTG_KLAUNCH((resstack_kernel<C, R, CEN>), dim3((unsigned)(a.B * a.n_t)), dim3(64 * kStackWaves), lds_bytes, s, a);
  return rtg_launch_status();
}

// which instance serves (C, L): 1 = the whole clip of 32 positions (C = 128), 2 / 3 = position tiles with halo (C = 64 / 32)
int stack_kind(const RtgResStackDesc* d) {
  if (d->C == 128 && d->L == 32) return 1;
  if (d->C == 64 && d->L >= 64) return 2;
  if (d->C == 32 && d->L >= 128) return 3;
  return 0;
}

int run(const RtgResStackDesc* d, const StackArgs& a, hipStream_t s) {
  switch (stack_kind(d)) {
    case 1: return launch<128, 32, 32>(a, d->L, s);
    case 2: return launch<64, 128, 96>(a, d->L, s);
    case 3: return launch<32, 256, 224>(a, d->L, s);
  }
  return RTG_EINVAL;
}

}  // namespace

extern "C" int rtg_resstack_ok(const RtgResStackDesc* d) {
  if (!d) return RTG_ENULL;
  if (d->B < 1 || d->B > 65535) return 0;
  for (int i = 0; i < 6; ++i)
    if (d->dil[i] < 1 || d->dil[i] > kHalo) return 0;
  if ((long long)d->B * ((d->L + 95) / 96) > (1 << 24)) return 0;
  // -> the instance that serves the shape (1: (128, 32), 2: (64, L >= 64), 3: (32, L >= 128)), 0: none.  Which instances a
  // caller USES is its choice (rtg/ops.py: RTG_RESSTACK_KINDS, default instance 1 only).  Measured at batch 32 inside the
  // train step (us per stack, forward / backward, against six launches of the general kernel): (128, 32) 73 / 115 vs 150 /
  // 170; (64, 256) 79 / 113 vs 102 / 120 and (32, 2048) 93 / 130 vs 96 / 114 (stand-alone) — no gain: a block per clip
  // (tile) leaves most of the chip idle or, tiled, moves 58-108 MB through 4-byte epilogue accesses.
  const int kind = stack_kind(d);
  return kind > 0 ? kind : 0;
}

extern "C" int rtg_resstack_forward(const RtgResStackDesc* d, const float* x, const float* const* wp,
                                    const float* const* bias, float* const* outs, void* stream) {
  if (!d || !x || !wp || !bias || !outs) return RTG_ENULL;
  if (rtg_resstack_ok(d) < 1) return RTG_EINVAL;
  StackArgs a;
  a.in = x; a.pro_aux = nullptr;
  for (int i = 0; i < 6; ++i) {
    if (!wp[i] || !outs[i]) return RTG_ENULL;
    a.wp[i] = wp[i]; a.bias[i] = bias[i]; a.mask[i] = nullptr; a.gout[i] = outs[i]; a.dil[i] = d->dil[i];
  }
  a.B = d->B; a.pre_slope = d->pre_slope; a.mask_slope = 1.f; a.final_act = d->final_act; a.act_slope = d->act_slope;
  return run(d, a, (hipStream_t)stream);
}

extern "C" int rtg_resstack_backward(const RtgResStackDesc* d, const float* dy, const float* y, const float* const* wpb,
                                     const float* const* masks, float* const* gouts, void* stream) {
  if (!d || !dy || !wpb || !masks || !gouts) return RTG_ENULL;
  if (rtg_resstack_ok(d) < 1) return RTG_EINVAL;
  if (d->final_act && !y) return RTG_ENULL;
  StackArgs a;
  a.in = dy; a.pro_aux = d->final_act ? y : nullptr;
  for (int i = 0; i < 6; ++i) {
    if (!wpb[i] || !masks[i] || !gouts[i]) return RTG_ENULL;
    a.wp[i] = wpb[i]; a.bias[i] = nullptr; a.mask[i] = masks[i]; a.gout[i] = gouts[i]; a.dil[i] = d->dil[i];
  }
  a.B = d->B; a.pre_slope = 1.f; a.mask_slope = d->pre_slope; a.final_act = d->final_act; a.act_slope = d->act_slope;
  return run(d, a, (hipStream_t)stream);
}
