// rtg_reschain.hip — a whole ResBlock3 branch of the UNet-G decoder (retunegan/models/generator.py:133-155, dilations
// generator.py:709-711) in ONE launch per direction, for the 32-channel stage (8192 positions per clip):
//
//   forward    for d in (9, 3, 1):  x <- x + conv_d(lrelu(x, 0.15)) + bias
//   backward   for d in (1, 3, 9):  g <- g + lrelu'(x_prev) * conv_d^T(g)
//
// — the same recurrence both ways: cur <- cur + M * (conv(act(cur)) + bias) with (act, M, bias) = (leaky-relu, 1, bias) forward
// and (identity, the leaky-relu derivative of the saved forward tensor, none) backward, the backward on the flipped
// (RTG_PACK_DGRAD_S1) weights.  Round 6, asked for since round 2.  The unfused path runs these layers as nine launches of
// rtg_resconv.hip per direction: short reductions (96 .. 224 deep), every launch pays its own prologue (the input window
// through registers into LDS) and epilogue (the result through LDS into 16-byte stores), 45-70 TFLOP/s.  Here
//   * a block owns a REGION of one clip — CEN central positions plus the chain's receptive field H = (k - 1) / 2 * (9 + 3 + 1)
//     on each side — in ONE LDS buffer, updated IN PLACE layer by layer: every layer is computed on the whole region (what the
//     halo misses only reaches H positions inward), all waves read, barrier, all waves add their tiles to the buffer, barrier;
//     positions outside the clip are forced to zero after every layer (each layer's own zero padding);
//   * the layer's result goes to HBM as well — the intermediates are the backward's masks and the weight gradients' operands
//     either way (forward: x1, x2, x3; backward: G2, G1, dx), so the fused chain saves the re-READS: per branch 1 map read + 3
//     written instead of 3 + 3 forward, 4 + 3 instead of 9 + 3 backward;
//   * v_mfma_f32_16x16x4_f32 with BOTH operands from LDS: the weights of the layer (12-28 KB) are copied into LDS once per
//     block and layer, rearranged on the way into [step][row tile][lane] so that a wave's A operand is one conflict-free
//     4-byte read; the next layer's weights travel through 6-14 registers per thread during this layer's multiplications.  No
//     A fragments in registers (rtg_resconv.hip keeps 48-112 per lane): 8-wave blocks at <= 128 registers, two blocks per CU —
//     one block's prologue, barriers and epilogues hide behind the other's matrix instructions;
//   * a wave owns all 32 rows of a contiguous run of 16-column tiles: per reduction step 2 A reads + one B read per column tile
//     feed 2 matrix instructions per column tile.
// The accumulation order of every output element — 16-channel chunk, tap, channel ascending, one fused multiply-add each
// (v_mfma_f32_16x16x4_f32 chains its four k like two v_mfma_f32_32x32x2_f32: rtg_dconv.hip) — and the epilogue arithmetic
// (fma(acc + bias, M, cur)) are the unfused kernels': results are bit-identical to the nine launches
// (tests/test_reschain_gpu.py).
#include "rtg_common.h"

namespace {

using rsrc_t = __amdgpu_buffer_rsrc_t;
#define RC_OOB 0x80000000u
constexpr int kC = 32;                 // channels (rows of every layer's operator)
constexpr int kWaves = 8;
constexpr int kMaxLayers = 3;
constexpr int kMaxTilesPerWave = 5;    // most 16-column tiles a wave owns (40 accumulator registers)

struct ChainArgs {
  const float* in;                     // x (forward) or dy (backward): [B, 32, L]
  const float* wp[kMaxLayers];         // standard packed image of each layer's operator (rtg_conv1d's: 32-row tiles)
  const float* bias[kMaxLayers];       // forward only
  const float* mask[kMaxLayers];       // backward only: the forward tensor whose sign masks this layer's result
  float* out[kMaxLayers];              // every layer's result in HBM
  int dil[kMaxLayers];
  int n_layers;
  int B, L, n_t, CEN, H;               // clips, clip length, tiles per clip, central positions per tile, halo per side
  int R16, q;                          // 16-column tiles of the region, tiles per wave
  int GL, Wp;                          // guard columns left of region column 0, floats between channel rows (== 16 mod 32)
  float pre_slope, mask_slope;
  int in_bytes;
  int dbg;
};

__device__ __forceinline__ f32x4 rc_load4(rsrc_t r, unsigned off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
}
__device__ __forceinline__ float rc_load(rsrc_t r, unsigned off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
}

// K taps; Q: 16-column tiles per wave; BWD: the leaky-relu-derivative mask in the epilogue (backward).
// LDS: act[32][Wp] | wbuf[K * 2 (chunks) * 4 (kq) * 2 (row tiles) * 64]
template <int K, int Q, bool BWD>
__global__ __launch_bounds__(kWaves * 64, 2) void reschain_kernel(const ChainArgs a) {
  constexpr int NS = 2 * K * 4;                      // reduction steps of a layer: (chunk, tap, kq), 4 channels each
  constexpr int NWF = NS * 2 * 64;                   // floats of a layer's weights
  constexpr int NTHR = kWaves * 64;
  constexpr int WPT = (NWF + NTHR - 1) / NTHR;       // weight floats a thread carries to LDS per layer
  constexpr int C0 = (K - 1) / 2;                    // centre tap
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* act = lds;
  float* wbuf = lds + kC * a.Wp;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n16 = lane & 15, kgrp = lane >> 4;
  const int b = blockIdx.x / a.n_t, ti = blockIdx.x - b * a.n_t;
  const int t0 = ti * a.CEN;                         // first central position
  const int rs = t0 - a.H;                           // clip position of region column 0
  const int R = a.R16 * 16;

  // ---- a layer's weights: standard image [(chunk, tap), channel pair][kk 2][row 32] -> wbuf[(chunk, tap, kq)][row tile][k 4][row 16]
  float wreg[WPT];
  auto w_issue = [&](int l) __attribute__((always_inline)) {
    const float* src = a.wp[l < a.n_layers ? l : a.n_layers - 1];
#pragma unroll
    for (int u = 0; u < WPT; ++u) {
      const int i = tid + u * NTHR;
      wreg[u] = i < NWF ? src[i] : 0.f;
    }
  };
  auto w_commit = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < WPT; ++u) {
      const int i = tid + u * NTHR;
      if (i < NWF) {
        const int f = i >> 6, kk = (i >> 5) & 1, m = i & 31;             // fragment (chunk, tap, pair), channel of the pair, row
        const int ct_ = f >> 3, cp = f & 7;                                // (chunk * K + tap), channel pair of the chunk
        const int kq = cp >> 1, k = ((cp & 1) << 1) | kk;                  // channel 2 cp + kk = 4 kq + k
        wbuf[((ct_ * 4 + kq) * 2 + (m >> 4)) * 64 + k * 16 + (m & 15)] = wreg[u];
      }
    }
  };
  w_issue(0);

  // ---- the buffer: zeros (guards, positions outside the clip), then the region of the clip
  for (int i = tid; i < kC * a.Wp / 4; i += NTHR) reinterpret_cast<f32x4*>(act)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  {
    const rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, a.in_bytes, 0x00020000);
    const int nq = a.R16 * 4;                         // quads per channel row
    for (int i = tid; i < kC * nq; i += NTHR) {
      const int ch = i / nq, j = i - ch * nq;
      const int t = rs + 4 * j;                       // clip position of the quad's first element
      if (t + 3 < 0 || t >= a.L) continue;
      const int e0 = (b * kC + ch) * a.L + t;         // element offset (negative only in front of the tensor's first row)
      f32x4 v;
      if (e0 >= 0) {
        v = rc_load4(rin, (unsigned)e0 * 4u);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = e0 + e >= 0 ? rc_load(rin, (unsigned)(e0 + e) * 4u) : 0.f;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (t + e < 0 || t + e >= a.L) v[e] = 0.f;
      *reinterpret_cast<f32x4*>(act + ch * a.Wp + a.GL + 4 * j) = v;
    }
  }
  w_commit();
  __syncthreads();

  // this wave's tiles: [c_lo, c_lo + nct) of the region's R16 column tiles
  const int c_lo = wave * Q;
  int nct = a.R16 - c_lo;
  nct = nct < 0 ? 0 : (nct > Q ? Q : nct);

  for (int l = 0; l < a.n_layers; ++l) {
    w_issue(l + 1);                                    // the next layer's weights travel during this layer's multiplications
    const int dil = a.dil[l];
    // backward: the mask operand of this wave's elements, requested before the multiplications
    [[maybe_unused]] float mk[BWD ? Q : 1][2][4];
    if constexpr (BWD) {
      const rsrc_t rmk = __builtin_amdgcn_make_buffer_rsrc((void*)a.mask[l], 0, a.in_bytes, 0x00020000);
#pragma unroll
      for (int c = 0; c < Q; ++c)
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int col = (c_lo + c) * 16 + n16, t = rs + col;
            const int m = rt * 16 + kgrp * 4 + r;
            const bool ok = c < nct && t >= 0 && t < a.L;
            mk[c][rt][r] = rc_load(rmk, ok ? (unsigned)((b * kC + m) * a.L + t) * 4u : RC_OOB);
          }
    }

    f32x4 acc[Q][2];
#pragma unroll
    for (int c = 0; c < Q; ++c) acc[c][0] = acc[c][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* bp = act + kgrp * a.Wp + a.GL + c_lo * 16 + n16;      // B operand of step (0, centre tap, 0), tile 0
    const float* ap = wbuf + lane;
    // one reduction step ahead: A (2 row tiles) and B (this wave's column tiles) of step s + 1 are read before step s multiplies
    float a_cur[2], a_nxt[2], b_cur[Q], b_nxt[Q];
    auto fetch = [&](int s, float (&av)[2], float (&bv)[Q]) __attribute__((always_inline)) {
      const int cc = s / (K * 4), tap = (s >> 2) % K, kq = s & 3;
      av[0] = ap[(s * 2 + 0) * 64];
      av[1] = ap[(s * 2 + 1) * 64];
      const float* brow = bp + (cc * 16 + kq * 4) * a.Wp + (tap - C0) * dil;
#pragma unroll
      for (int c = 0; c < Q; ++c) bv[c] = brow[c * 16];     // (tiles past nct read the neighbour's columns or the guard: unused)
    };
    fetch(0, a_cur, b_cur);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      if (a.dbg & 4) break;
      float(&ac)[2] = (s & 1) ? a_nxt : a_cur;
      float(&an)[2] = (s & 1) ? a_cur : a_nxt;
      float(&bc)[Q] = (s & 1) ? b_nxt : b_cur;
      float(&bn)[Q] = (s & 1) ? b_cur : b_nxt;
      if (s + 1 < NS) fetch(s + 1, an, bn);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c = 0; c < Q; ++c) {
        float v = bc[c];
        v = v > 0.f ? v : v * a.pre_slope;
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) acc[c][rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[rt], v, acc[c][rt], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();                                   // every wave is done reading the buffer and the weights
    if (a.dbg & 2) { w_commit(); __syncthreads(); continue; }
    // ---- epilogue: cur <- cur + M * (acc + bias), zero outside the clip, in place; the central positions to HBM
    {
      float* outp = a.out[l];
      const float* bias = a.bias[l];
#pragma unroll
      for (int c = 0; c < Q; ++c) {
        if (c >= nct) continue;
        const int col = (c_lo + c) * 16 + n16, t = rs + col;
        const bool inside = t >= 0 && t < a.L;
        const bool central = inside && col >= a.H && col < a.H + a.CEN;
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int m = rt * 16 + kgrp * 4 + r;
            float* pc = act + m * a.Wp + a.GL + col;
            const float cur = *pc;
            float v = acc[c][rt][r] + (bias ? bias[m] : 0.f);
            float mf = 1.f;
            if constexpr (BWD) mf = mk[c][rt][r] > 0.f ? 1.f : a.mask_slope;
            v = __builtin_fmaf(v, mf, cur);
            *pc = inside ? v : 0.f;
            if (central && !(a.dbg & 1)) outp[(size_t)(b * kC + m) * a.L + t] = v;
          }
      }
    }
    w_commit();
    __syncthreads();
  }
}

template <int K, int Q, bool BWD>
int launch(const ChainArgs& a, size_t lds_bytes, hipStream_t s) {
  auto k = reschain_kernel<K, Q, BWD>;
  static std::atomic<unsigned> optin{0};              // (> 64 KB of dynamic LDS: opt-in per kernel and device)
  if (lds_bytes > 64 * 1024 && rtg_lds_optin((const void*)k, optin) != RTG_OK) return RTG_ERANGE;
  RTG_KLAUNCH(k, dim3((unsigned)(a.B * a.n_t)), dim3(kWaves * 64), lds_bytes, s, a);
  return rtg_launch_status();
}

// geometry of the launch for descriptor d: the region (in 16-column tiles, a multiple-of-8-friendly count for the 8 waves) whose
// block fits half a CU's LDS (two blocks per CU) and costs least: (rounds of the chip at two blocks per CU) x (tiles per wave)
int geometry(const RtgResChainDesc* d, ChainArgs* a, size_t* lds_bytes) {
  if (d->C != kC || (d->K != 3 && d->K != 5 && d->K != 7) || d->n_layers < 1 || d->n_layers > kMaxLayers) return RTG_EINVAL;
  if (d->B < 1 || d->L < 64 || (d->L & 3) != 0 || (long long)d->B * kC * d->L * 4 >= (1ll << 31)) return RTG_ERANGE;
  int H = 0, dmax = 0;
  for (int i = 0; i < d->n_layers; ++i) {
    if (d->dil[i] < 1 || d->dil[i] > 9) return RTG_EINVAL;
    H += (d->K - 1) / 2 * d->dil[i];
    dmax = d->dil[i] > dmax ? d->dil[i] : dmax;
  }
  const int NWF = 2 * d->K * 4 * 2 * 64;
  const int GL = ((d->K - 1) / 2 * dmax + 3) & ~3;
  long long best_cost = -1;
  int best = 0;
  const int force = RTG_ENV_INT("RTG_DEV_RC_R16", 0);
  const size_t lds_cap = (size_t)RTG_ENV_INT("RTG_DEV_RC_LDS", 80) * 1024;
  for (int r16 = kWaves * kMaxTilesPerWave; r16 >= 8; --r16) {
    if (force && r16 != force) continue;
    const int R = r16 * 16;
    int Wp = GL + R + GL;
    Wp += (16 - (Wp & 31) + 32) & 31;                                       // == 16 (mod 32)
    const size_t bytes = ((size_t)kC * Wp + NWF) * sizeof(float);
    const int cen = (R - 2 * H) & ~3;
    if (bytes > lds_cap || cen < 64) continue;
    const int q = rtg_ceil_div(r16, kWaves);
    if (q < 2) continue;
    const long long blocks = (long long)d->B * rtg_ceil_div(d->L, cen);
    const long long cost = ((blocks + 511) / 512) * q * 1000 + (kWaves * q - r16);      // (ties: the fewest idle tile slots)
    if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = r16; a->Wp = Wp; *lds_bytes = bytes; }
  }
  if (!best) return RTG_ERANGE;
  a->GL = GL;
  a->R16 = best;
  a->H = H;
  a->q = rtg_ceil_div(best, kWaves);
  a->n_t = rtg_ceil_div(d->L, (best * 16 - 2 * H) & ~3);
  // even tiles: every tile of a clip as wide as the others (fewer, equal tiles beat a short last one)
  a->CEN = (rtg_ceil_div(d->L, a->n_t) + 3) & ~3;
  if ((long long)d->B * a->n_t > (1 << 24)) return RTG_ERANGE;
  a->B = d->B; a->L = d->L; a->n_layers = d->n_layers;
  a->in_bytes = d->B * kC * d->L * 4;
  a->dbg = RTG_ENV_INT("RTG_DEV_RC_DBG", 0);
  return RTG_OK;
}

template <int K, bool BWD>
int launch_q(const ChainArgs& a, size_t lds_bytes, hipStream_t s) {
  switch (a.q) {
    case 2: return launch<K, 2, BWD>(a, lds_bytes, s);
    case 3: return launch<K, 3, BWD>(a, lds_bytes, s);
    case 4: return launch<K, 4, BWD>(a, lds_bytes, s);
    case 5: return launch<K, 5, BWD>(a, lds_bytes, s);
  }
  return RTG_ERANGE;
}

template <bool BWD>
int run(const RtgResChainDesc* d, ChainArgs& a, hipStream_t s) {
  size_t lds_bytes = 0;
  const int st = geometry(d, &a, &lds_bytes);
  if (st) return st;
  switch (d->K) {
    case 3: return launch_q<3, BWD>(a, lds_bytes, s);
    case 5: return launch_q<5, BWD>(a, lds_bytes, s);
    default: return launch_q<7, BWD>(a, lds_bytes, s);
  }
}

}  // namespace

extern "C" int rtg_reschain_ok(const RtgResChainDesc* d) {
  if (!d) return RTG_ENULL;
  ChainArgs a;
  size_t lds = 0;
  return geometry(d, &a, &lds) == RTG_OK ? 1 : 0;
}

extern "C" int rtg_reschain_forward(const RtgResChainDesc* d, const float* x, const float* const* wp, const float* const* bias,
                                    float* const* outs, void* stream) {
  if (!d || !x || !wp || !bias || !outs) return RTG_ENULL;
  ChainArgs a;
  a.in = x;
  for (int i = 0; i < kMaxLayers; ++i) {
    const bool live = i < d->n_layers;
    if (live && (!wp[i] || !outs[i])) return RTG_ENULL;
    a.wp[i] = live ? wp[i] : nullptr; a.bias[i] = live ? bias[i] : nullptr; a.mask[i] = nullptr;
    a.out[i] = live ? outs[i] : nullptr; a.dil[i] = live ? d->dil[i] : 1;
  }
  a.pre_slope = d->pre_slope; a.mask_slope = 1.f;
  return run<false>(d, a, (hipStream_t)stream);
}

extern "C" int rtg_reschain_backward(const RtgResChainDesc* d, const float* dy, const float* const* wpb, const float* const* masks,
                                     float* const* gouts, void* stream) {
  if (!d || !dy || !wpb || !masks || !gouts) return RTG_ENULL;
  ChainArgs a;
  a.in = dy;
  for (int i = 0; i < kMaxLayers; ++i) {
    const bool live = i < d->n_layers;
    if (live && (!wpb[i] || !masks[i] || !gouts[i])) return RTG_ENULL;
    a.wp[i] = live ? wpb[i] : nullptr; a.bias[i] = nullptr; a.mask[i] = live ? masks[i] : nullptr;
    a.out[i] = live ? gouts[i] : nullptr; a.dil[i] = live ? d->dil[i] : 1;
  }
  a.pre_slope = 1.f; a.mask_slope = d->pre_slope;
  return run<true>(d, a, (hipStream_t)stream);
}
