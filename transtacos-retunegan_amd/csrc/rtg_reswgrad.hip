// rtg_reswgrad.hip — weight / bias gradients of the stride-1 "same" convolutions of the UNet-G residual blocks
// (ResBlock3 / ResidualStack, C_in = C_out in {32, 64}, k in {3, 5, 7}: generator.py:33-77,133-155) as a streaming
// reduction on the fp32 matrix cores.
//
//   dW[m][c][j] = sum_{b,t} gy[b,m,t] * pre(x[b,c,t - pad + j*dil]),   db[m] = sum_{b,t} gy[b,m,t]
// The output is tiny (C x (C*k + 1) floats) and both operands are read once: 67 MB for 1.6 .. 3.8 GFLOP at batch 32, i.e.
// 10 .. 24 us at the chip's rates, where the general kernel (rtg_wgrad_kernel.h: column = (channel, tap) pairs of a channel
// chunk, many small blocks with split-K partials) takes 48 .. 66 us.  Here
//   * a block owns a run of position tiles and keeps the WHOLE dW of its row tile in registers: accumulators
//     [channel block][tap] of 32x32 MFMA tiles (k7: 7 sets at C = 32, 14 at C = 64); the reduction index of
//     v_mfma_f32_32x32x2_f32 is the position: A = gy[m][t .. t+1], B = x[c][t + j*dil - pad .. +1] for the tile (c-block, j);
//   * the waves of a block split the POSITIONS of a tile (and, at C = 64, the two row tiles), so one A fragment feeds every
//     (channel block, tap) tile: 1 + C/32 * k LDS reads per C/32 * k MFMAs;
//   * both operand tiles are staged raw through registers while the previous tile is multiplied (odd LDS row pitches: the
//     fragment reads run across channels / rows at one position, conflict-free);
//   * the waves' sums meet in LDS in fixed order (bitwise reproducible) and leave as ONE split partial per block in the
//     layout rtg_weightnorm_backward reduces.
// Exposed as shape code 8 of RtgWgradDesc.shape_cfg (a tuner candidate next to the general shapes).
#include "rtg_common.h"

namespace {

struct RwArgs {
  const float *x, *dy;
  float* part;
  long long part_stride;
  int B, L, dil, pad, m, NQx, Wp, Dp, n_t, items;
  int pre;
  float pre_slope, gy_scale;
};

__device__ __forceinline__ int mrow32(int lane, int r) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

template <int CIN, int KT>
__global__ __launch_bounds__(RTG_THREADS, 1) void reswgrad_kernel(const RwArgs a) {
  constexpr int NRT = CIN / 32;                  // row tiles (= channel blocks)
  constexpr int NPW = 4 / NRT;                   // waves along the positions per row tile
  constexpr int PT = CIN == 32 ? 256 : 128;      // positions per tile
  constexpr int PWAVE = PT / NPW;                // positions per wave and tile
  constexpr int NACC = NRT * KT;                 // MFMA tiles per wave: (channel block, tap)
  constexpr int TPROW = PT / 2, RP = RTG_THREADS / TPROW, NLD = CIN / RP;
  constexpr int INNER = CIN * KT;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* xL = lds;                               // [CIN][Wp]
  float* dL = lds + CIN * a.Wp;                  // [CIN][Dp]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rt = wave % NRT, pw = wave / NRT;
  const int kk = lane >> 5, n_lane = lane & 31;

  const int per = (a.items + (int)gridDim.x - 1) / (int)gridDim.x;
  const int lo = blockIdx.x * per, hi = min(a.items, lo + per);

  // ---- staging (thread = float4 column q of rows rr, rr + RP, ...): x window from a0 = floor4(t0 - pad), gy tile from t0
  const int q = tid & (TPROW - 1), rr = tid / TPROW;
  f32x4 px[NLD], pd[NLD];
  auto issue = [&](int item) __attribute__((always_inline)) {
    const int b = item / a.n_t, t0 = (item - b * a.n_t) * PT;
    const int pos = ((t0 - a.pad) & ~3) + 4 * q;
    const bool okx = q < a.NQx && pos >= 0 && pos < a.L;
    const bool okd = 4 * q < PT && t0 + 4 * q < a.L;
    const float* sx = a.x + ((size_t)b * CIN + rr) * a.L + pos;
    const float* sd = a.dy + ((size_t)b * CIN + rr) * a.L + t0 + 4 * q;
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      px[u] = okx ? *reinterpret_cast<const f32x4*>(sx + (size_t)u * RP * a.L) : f32x4{0.f, 0.f, 0.f, 0.f};
      pd[u] = okd ? *reinterpret_cast<const f32x4*>(sd + (size_t)u * RP * a.L) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  const float pslope = a.pre ? a.pre_slope : 1.f;
  auto commit = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      if (q < a.NQx) {
        float* d_ = xL + (rr + u * RP) * a.Wp + 4 * q;
        const float t4[4] = {px[u].x, px[u].y, px[u].z, px[u].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) d_[e] = t4[e] > 0.f ? t4[e] : t4[e] * pslope;
      }
      if (4 * q < PT) {
        float* d_ = dL + (rr + u * RP) * a.Dp + 4 * q;
        d_[0] = pd[u].x; d_[1] = pd[u].y; d_[2] = pd[u].z; d_[3] = pd[u].w;
      }
    }
  };

  f32x16 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float bsum = 0.f;

  if (lo < hi) issue(lo);
  const float* ap = dL + (rt * 32 + n_lane) * a.Dp + pw * PWAVE + kk;            // A: gy[row][position]
  const float* bp = xL + n_lane * a.Wp + a.m + pw * PWAVE + kk;                   // B: x[channel][position + tap * dil]
  for (int item = lo; item < hi; ++item) {
    __syncthreads();                                   // the previous tile is consumed
    commit();
    __syncthreads();
    if (item + 1 < hi) issue(item + 1);
    constexpr int NS = PWAVE / 2, PD = NACC > 10 ? 1 : 2;   // fragment reads requested PD steps ahead of their MFMAs
    float va[PD + 1], vb[PD + 1][NACC];
    auto fload = [&](int s_, float& fa, float (&fb)[NACC]) __attribute__((always_inline)) {
      fa = ap[2 * s_];
#pragma unroll
      for (int cb = 0; cb < NRT; ++cb)
#pragma unroll
        for (int j = 0; j < KT; ++j) fb[cb * KT + j] = bp[cb * 32 * a.Wp + 2 * s_ + j * a.dil];
    };
#pragma unroll
    for (int s_ = 0; s_ < PD; ++s_) fload(s_, va[s_], vb[s_]);
#pragma unroll
    for (int s_ = 0; s_ < NS; ++s_) {
      if (s_ + PD < NS) fload(s_ + PD, va[(s_ + PD) % (PD + 1)], vb[(s_ + PD) % (PD + 1)]);
      __builtin_amdgcn_sched_barrier(0);
      const float av = va[s_ % (PD + 1)];
      bsum += av;
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, vb[s_ % (PD + 1)][i], acc[i], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- the waves' sums meet in LDS, position-wave 0 first (fixed order), then leave as one partial of this block:
  // o[row][INNER + 1] (bias last), copied out in the layout [rows][INNER], [rows]
  __syncthreads();
  float* o = lds;
  for (int ph = 0; ph < NPW; ++ph) {
    if (pw == ph) {
#pragma unroll
      for (int cb = 0; cb < NRT; ++cb)
#pragma unroll
        for (int j = 0; j < KT; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float* p = o + (rt * 32 + mrow32(lane, r)) * (INNER + 1) + (cb * 32 + n_lane) * KT + j;
            *p = ph == 0 ? acc[cb * KT + j][r] : *p + acc[cb * KT + j][r];
          }
      const float bs = bsum + __shfl_down(bsum, 32, 64);
      if (lane < 32) {
        float* p = o + (rt * 32 + n_lane) * (INNER + 1) + INNER;
        *p = ph == 0 ? bs : *p + bs;
      }
    }
    __syncthreads();
  }
  float* part = a.part + (size_t)blockIdx.x * a.part_stride;
  for (int idx = tid; idx < CIN * INNER; idx += RTG_THREADS) {
    const int row = idx / INNER, col = idx - row * INNER;
    part[idx] = o[row * (INNER + 1) + col] * a.gy_scale;
  }
  for (int row = tid; row < CIN; row += RTG_THREADS) part[CIN * INNER + row] = o[row * (INNER + 1) + INNER] * a.gy_scale;
}

template <int CIN, int KT>
int launch(const RwArgs& a, int splits, size_t lds_bytes, hipStream_t s) {
  static std::atomic<unsigned> optin{0};              // (> 64 KB of dynamic LDS: opt-in per kernel and device)
  if (rtg_lds_optin(reinterpret_cast<const void*>(&reswgrad_kernel<CIN, KT>), optin) != RTG_OK) return RTG_ERANGE;
  RTG_KLAUNCH((reswgrad_kernel<CIN, KT>), dim3(splits), dim3(RTG_THREADS), lds_bytes, s, a);
  return rtg_launch_status();
}

}  // namespace

int rtg_reswgrad_ok(const RtgWgradDesc* d) {
  if (d->groups != 1 || d->C2 != 0 || d->stride != 1 || d->h_k > 1 || d->h_n > 1) return 0;
  if (d->Cg != d->Mg || (d->Cg != 32 && d->Cg != 64) || d->C1 != d->Cg) return 0;
  if (d->K != 3 && d->K != 5 && d->K != 7) return 0;
  if (d->Cg == 64 && d->K == 7) return 0;            // 14 accumulator tiles + the staging registers spill (86 vs 55 us)
  if (d->Q != d->L_in || d->dy_L != d->L_in || d->L_in % 4 != 0 || d->L_in < 128) return 0;
  if (d->pad < 0 || d->pad > (d->K - 1) * d->dil || d->dil > 16) return 0;
  if (d->pre_mode != RTG_PRE_NONE && d->pre_mode != RTG_PRE_LRELU) return 0;
  if (d->gy_mode != RTG_PRE_NONE) return 0;
  return 1;
}

int rtg_reswgrad_splits(const RtgWgradDesc* d) {
  if (!rtg_reswgrad_ok(d)) return RTG_EINVAL;
  const int PT = d->Cg == 32 ? 256 : 128;
  const long long items = (long long)d->B * rtg_ceil_div(d->L_in, PT);
  return (int)(items < 256 ? items : 256);
}

int rtg_reswgrad_launch(const RtgWgradDesc* d, const float* x, const float* dy, float* part, hipStream_t s) {
  if (!rtg_reswgrad_ok(d)) return RTG_EINVAL;
  if (d->splits != rtg_reswgrad_splits(d)) return RTG_EINVAL;
  if ((reinterpret_cast<uintptr_t>(x) & 15) != 0 || (reinterpret_cast<uintptr_t>(dy) & 15) != 0) return RTG_EINVAL;
  RwArgs a;
  a.x = x; a.dy = dy; a.part = part; a.part_stride = d->part_stride;
  a.B = d->B; a.L = d->L_in; a.dil = d->dil; a.pad = d->pad;
  const int PT = d->Cg == 32 ? 256 : 128;
  a.m = (-d->pad) & 3;
  const int w = a.m + PT + (d->K - 1) * d->dil;
  a.NQx = (w + 3) / 4;
  if (a.NQx > PT / 2) return RTG_ERANGE;
  a.Wp = (4 * a.NQx) | 1;                              // odd pitches: fragment reads run across rows at one position
  a.Dp = PT | 1;
  a.n_t = rtg_ceil_div(d->L_in, PT);
  a.items = d->B * a.n_t;
  a.pre = d->pre_mode == RTG_PRE_LRELU ? 1 : 0;
  a.pre_slope = d->pre_slope;
  a.gy_scale = d->gy_scale;
  const size_t stage = (size_t)d->Cg * (a.Wp + a.Dp);
  const size_t outp = (size_t)d->Cg * (d->Cg * d->K + 1);
  const size_t lds_bytes = (stage > outp ? stage : outp) * sizeof(float);
  if (lds_bytes > 160 * 1024) return RTG_ERANGE;
#define RTG_RW(c, k) \
  if (d->Cg == c && d->K == k) return launch<c, k>(a, d->splits, lds_bytes, s);
  RTG_RW(32, 3) RTG_RW(32, 5) RTG_RW(32, 7) RTG_RW(64, 3) RTG_RW(64, 5)
#undef RTG_RW
  return RTG_EINVAL;
}
