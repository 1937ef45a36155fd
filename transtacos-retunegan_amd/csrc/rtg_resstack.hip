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
  int dbg;
};

__device__ __forceinline__ int mrow32(int lane, int r) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

constexpr int kStackWaves = 8;
constexpr int kHalo = 9;

template <int C, int L>
__global__ __launch_bounds__(64 * kStackWaves) void resstack_kernel(const StackArgs a) {
  constexpr int NCC = C / RTG_CK;                 // 16-channel chunks
  constexpr int RT = C / 32;                      // row tiles
  constexpr int NKS = C == 128 ? 2 : 1;           // K splits across waves
  constexpr int CHP = NCC / NKS;                  // chunks per wave: 4
  constexpr int NT = C == 128 ? 1 : 2;            // column tiles per wave
  constexpr int NA = CHP * 3 * 8;                 // A fragments per wave and layer: 96
  constexpr int LP = L + 2 * kHalo + 2;           // LDS row pitch (zero halo on both sides)
  static_assert(CHP == 4 && RT * NKS * (L / 32 / NT) == kStackWaves, "wave decomposition");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* bufA = lds;
  float* bufB = lds + C * LP;
  float* scr = lds + 2 * C * LP;                  // K-split meeting point (C = 128): [RT][32 x 32]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kk = lane >> 5, n_lane = lane & 31;
  const int rt = wave % RT;
  const int ks = C == 128 ? wave / RT : 0;
  const int ct0 = C == 128 ? 0 : (wave / RT) * NT;
  const int b = blockIdx.x;

  // ---- A fragments of a layer: ((chunk, tap), channel pair) -> 64 consecutive floats in the packed layout
  // (C = 128: two register sets, the next layer's fragments are requested a layer ahead; C = 64 has two accumulator
  // tiles per wave and no room for the second set: its fragments are requested at the top of the layer and waited for one
  // by one as the MFMA loop reaches them)
  constexpr bool DB = C == 128;
  float A0[NA], A1[DB ? NA : 1];
  auto aload = [&](int l, float (&Af)[NA]) __attribute__((always_inline)) {
    const float* w = a.wp[l] + ((size_t)(rt * NCC + ks * CHP) * 3) * 8 * 64 + lane;
#pragma unroll
    for (int f = 0; f < NA; ++f) Af[f] = w[f * 64];
  };
  aload(0, A0);

  // ---- buffers: zero (the halos stay zero: every conv sees zero padding), then the clip
  for (int i = tid; i < 2 * C * LP; i += 64 * kStackWaves) lds[i] = 0.f;
  __syncthreads();
  for (int i = tid; i < C * L; i += 64 * kStackWaves) {
    const int c = i / L, t = i - c * L;
    const size_t gi = ((size_t)b * C + c) * L + t;
    float v = a.in[gi];
    if (a.pro_aux && a.final_act) v *= (a.pro_aux[gi] > 0.f ? 1.f : a.act_slope);
    bufA[c * LP + kHalo + t] = v;
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
    f32x16 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const float* bp = in + (ks * CHP * RTG_CK + kk) * LP + kHalo + ct0 * 32 + n_lane - dil;
    constexpr int NS = NA, PD = NT == 1 ? 4 : 2;    // B-fragment reads requested PD steps (>= 256 cycles of MFMAs) ahead
    float vb[PD + 1][NT];
    auto bload = [&](int s_, float (&dst)[NT]) __attribute__((always_inline)) {
      const int cci = s_ / 24, tap = (s_ / 8) % 3, cp = s_ % 8;
      const float* brow = bp + (cci * RTG_CK + cp * 2) * LP + tap * dil;
#pragma unroll
      for (int j = 0; j < NT; ++j) dst[j] = brow[j * 32];
    };
    if (!(a.dbg & 1)) {
#pragma unroll
    for (int s_ = 0; s_ < PD; ++s_) bload(s_, vb[s_]);
#pragma unroll
    for (int s_ = 0; s_ < NS; ++s_) {
      if (s_ + PD < NS) bload(s_ + PD, vb[(s_ + PD) % (PD + 1)]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        float v = vb[s_ % (PD + 1)][j];
        v = v > 0.f ? v : v * a.pre_slope;
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(Acur[s_], v, acc[j], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    }
    if (NKS == 2) {                                // the upper channel half hands its sums to the lower one
      if (ks == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) scr[rt * 1024 + r * 64 + lane] = acc[0][r];
      }
      __syncthreads();
      if (ks == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][r] += scr[rt * 1024 + r * 64 + lane];
      }
    }
    __syncthreads();                               // every wave is done reading `in` (and, odd layers, may overwrite bufA)
    if (ks == 0 && !(a.dbg & 2)) {
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int pos = (ct0 + j) * 32 + n_lane;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = rt * 32 + mrow32(lane, r);
          const size_t gi = ((size_t)b * C + m) * L + pos;
          float v = acc[j][r] + (a.bias[l] ? a.bias[l][m] : 0.f);
          const float mf = a.mask[l] ? (a.mask[l][gi] > 0.f ? 1.f : a.mask_slope) : 1.f;
          const float rv = odd ? outb[m * LP + kHalo + pos] : 0.f;
          v = __builtin_fmaf(v, mf, rv);
          if (l < 5) outb[m * LP + kHalo + pos] = v;
          if (l == 5 && a.final_act && !a.pro_aux) v = v > 0.f ? v : v * a.act_slope;
          a.gout[l][gi] = v;
        }
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

template <int C, int L>
int launch(const StackArgs& a, hipStream_t s) {
  constexpr int LP = L + 2 * kHalo + 2;
  const size_t lds_bytes = ((size_t)2 * C * LP + (C == 128 ? 4 * 1024 : 0)) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&resstack_kernel<C, L>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return RTG_ERANGE;
    attr_set = true;
  }
  RTG_KLAUNCH((resstack_kernel<C, L>), dim3(a.B), dim3(64 * kStackWaves), lds_bytes, s, a);
  return rtg_launch_status();
}

int run(const RtgResStackDesc* d, const StackArgs& a, hipStream_t s) {
  if (d->C == 128 && d->L == 32) return launch<128, 32>(a, s);
  if (d->C == 64 && d->L == 256) return launch<64, 256>(a, s);
  return RTG_EINVAL;
}

}  // namespace

extern "C" int rtg_resstack_ok(const RtgResStackDesc* d) {
  if (!d) return RTG_ENULL;
  if (d->B < 1 || d->B > 65535) return 0;
  for (int i = 0; i < 6; ++i)
    if (d->dil[i] < 1 || d->dil[i] > kHalo) return 0;
  // (C, L) = (64, 256) is built and tested (RTG_RESSTACK_ALL=1) but not served by default: one block per clip is 32 blocks
  // at batch 32, and its 1536 MFMAs per layer and block take longer on 32 CUs (126 / 166 us per stack and direction) than
  // six launches of the general kernel spread over the chip (102 / 120 us); (128, 32): 66 / 71 against 150 / 170 us
  if (d->C == 64 && d->L == 256) return getenv("RTG_RESSTACK_ALL") ? 1 : 0;
  return (d->C == 128 && d->L == 32) ? 1 : 0;
}

extern "C" int rtg_resstack_forward(const RtgResStackDesc* d, const float* x, const float* const* wp,
                                    const float* const* bias, float* const* outs, void* stream) {
  if (!d || !x || !wp || !bias || !outs) return RTG_ENULL;
  if (rtg_resstack_ok(d) != 1) return RTG_EINVAL;
  StackArgs a;
  a.in = x; a.pro_aux = nullptr;
  for (int i = 0; i < 6; ++i) {
    if (!wp[i] || !outs[i]) return RTG_ENULL;
    a.wp[i] = wp[i]; a.bias[i] = bias[i]; a.mask[i] = nullptr; a.gout[i] = outs[i]; a.dil[i] = d->dil[i];
  }
  a.B = d->B; a.pre_slope = d->pre_slope; a.mask_slope = 1.f; a.final_act = d->final_act; a.act_slope = d->act_slope;
  a.dbg = getenv("RTG_RS_DBG") ? atoi(getenv("RTG_RS_DBG")) : 0;
  return run(d, a, (hipStream_t)stream);
}

extern "C" int rtg_resstack_backward(const RtgResStackDesc* d, const float* dy, const float* y, const float* const* wpb,
                                     const float* const* masks, float* const* gouts, void* stream) {
  if (!d || !dy || !wpb || !masks || !gouts) return RTG_ENULL;
  if (rtg_resstack_ok(d) != 1) return RTG_EINVAL;
  if (d->final_act && !y) return RTG_ENULL;
  StackArgs a;
  a.in = dy; a.pro_aux = d->final_act ? y : nullptr;
  for (int i = 0; i < 6; ++i) {
    if (!wpb[i] || !masks[i] || !gouts[i]) return RTG_ENULL;
    a.wp[i] = wpb[i]; a.bias[i] = nullptr; a.mask[i] = masks[i]; a.gout[i] = gouts[i]; a.dil[i] = d->dil[i];
  }
  a.B = d->B; a.pre_slope = 1.f; a.mask_slope = d->pre_slope; a.final_act = d->final_act; a.act_slope = d->act_slope;
  a.dbg = getenv("RTG_RS_DBG") ? atoi(getenv("RTG_RS_DBG")) : 0;
  return run(d, a, (hipStream_t)stream);
}
