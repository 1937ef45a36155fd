// rtg_conv1d.hip — implicit-GEMM Conv1d on the gfx950 fp32 matrix cores.
//
// One kernel family serves: Conv1d forward (any stride / dilation / groups), ConvTranspose1d forward and every
// convolution backward-data of the RetuneGAN path (through repacked weights: rtg_weights.hip).  See include/rtg.h
// (rtg_conv1d) for the operator definition and the reference call sites it replaces.
//
// Mapping to the hardware:
//   GEMM view   M = output rows of one group, N = batch x output positions, K = (input channel, tap).
//   workgroup   256 threads = 4 wavefronts arranged WM x WN; wave tile = MT x NT MFMA tiles of TM x TM
//               (TM = 32: v_mfma_f32_32x32x2_f32, TM = 16: v_mfma_f32_16x16x4_f32 for 16-row groups).
//   B operand   the input patch of 16 channels x (BN-1)*stride+(K-1)*dil+1 samples is staged ONCE per block into LDS
//               (coalesced global loads, input activation fused at staging), double buffered across channel chunks;
//               every tap / every output row re-reads it from LDS with ds_read_b32 at lane-consecutive addresses
//               (strided convs de-interleave the patch by phase so that lanes stay bank-conflict free).
//   A operand   weights are pre-packed (rtg_weights_pack) so that one MFMA fragment is 64 consecutive floats:
//               each wave loads its fragments straight from L2 with one coalesced 256-B load, prefetched one
//               (chunk, tap) step ahead — no LDS traffic and no barrier for weights.
//   epilogue    bias, leaky-relu-derivative mask, residual, scale, activation, optional polyphase "shuffle" store.
#include "rtg_common.h"

namespace {

struct ConvArgs {
  const float *x1, *x2, *aux, *wp, *bias, *mask, *res;
  float *out, *out2;
  int out_split;
  int B, C1, C2, L_in, groups, Cg, Mg, K, stride, dil, pad, Q, out_C, out_L, shuf_S, shuf_P;
  int pre_mode;
  float pre_slope, mask_slope, out_scale;
  int act;
  float act_slope;
  int accumulate;
  int n_cc, n_mt, WM, WN, PW, PH, ROW, m_blocks;
};

template <int TM>
struct Mfma;
template <>
struct Mfma<32> {
  using acc_t = f32x16;
  static constexpr int NREG = 16;
  static __device__ __forceinline__ acc_t run(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int row(int lane, int r) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }
};
template <>
struct Mfma<16> {
  using acc_t = f32x4;
  static constexpr int NREG = 4;
  static __device__ __forceinline__ acc_t run(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int row(int lane, int r) { return (lane >> 4) * 4 + r; }
};

template <int TM, int MT, int NT>
__global__ __launch_bounds__(RTG_THREADS) void conv1d_mfma_kernel(const ConvArgs a) {
  using M = Mfma<TM>;
  using acc_t = typename M::acc_t;
  constexpr int KK = 64 / TM;           // K-values consumed per MFMA
  constexpr int CPN = RTG_CK / KK;      // MFMA k-steps per (chunk, tap)
  constexpr int MAXIT = RTG_PW_MAX / 64;
  constexpr int RPW = RTG_CK / 4;       // patch rows staged per wave

  extern __shared__ __attribute__((aligned(16))) float lds[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / a.WN, wn = wave - wm * a.WN;
  const int b = blockIdx.z;
  const int g = blockIdx.y / a.m_blocks, mb = blockIdx.y - g * a.m_blocks;
  const int mt0 = (mb * a.WM + wm) * MT;
  const int BN = a.WN * NT * TM;
  const int q_blk = blockIdx.x * BN;
  const int o_start = q_blk * a.stride - a.pad;
  const int bufsz = RTG_CK * a.ROW;

  // ---- staging geometry (per thread, independent of the channel chunk)
  int loff[MAXIT];
#pragma unroll
  for (int it = 0; it < MAXIT; ++it) {
    const int o = lane + 64 * it;
    loff[it] = (a.stride == 1) ? o : (o % a.stride) * a.PH + o / a.stride;
  }
  float st[RPW][MAXIT];

  auto gload = [&](int cc) {
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const int c = cc * RTG_CK + wave * RPW + i;
      const int gc = g * a.Cg + c;
      const bool cvalid = c < a.Cg;
      const float* src;
      const float* asrc = nullptr;
      if (gc < a.C1) {
        src = a.x1 + ((size_t)b * a.C1 + gc) * a.L_in;
        if (a.aux) asrc = a.aux + ((size_t)b * a.C1 + gc) * a.L_in;
      } else {
        src = a.x2 + ((size_t)b * a.C2 + (gc - a.C1)) * a.L_in;
      }
#pragma unroll
      for (int it = 0; it < MAXIT; ++it) {
        const int o = lane + 64 * it;
        const int pos = o_start + o;
        const bool ok = cvalid && o < a.PW && pos >= 0 && pos < a.L_in;
        float v = ok ? src[pos] : 0.f;
        if (a.pre_mode == RTG_PRE_LRELU) {
          v = rtg_lrelu(v, a.pre_slope);
        } else if (a.pre_mode == RTG_PRE_MUL_DLRELU) {
          const float av = ok ? asrc[pos] : 0.f;
          v = v * (av > 0.f ? 1.f : a.pre_slope);
        } else if (a.pre_mode == RTG_PRE_MUL_DTANH) {
          const float av = ok ? asrc[pos] : 0.f;
          v = v * (1.f - av * av);
        }
        st[i][it] = v;
      }
    }
  };
  auto swrite = [&](float* buf) {
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      float* rowp = buf + (wave * RPW + i) * a.ROW;
#pragma unroll
      for (int it = 0; it < MAXIT; ++it)
        if (lane + 64 * it < a.PW) rowp[loff[it]] = st[i][it];
    }
  };

  // ---- accumulators
  acc_t acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < M::NREG; ++r) acc[i][j][r] = 0.f;

  // ---- operand addressing
  const int n_lane = lane & (TM - 1), kk = lane / TM;
  const int bbase = kk * a.ROW + wn * NT * TM + n_lane;
  const float* wptr[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    int mt = mt0 + i;
    if (mt > a.n_mt - 1) mt = a.n_mt - 1;   // clamped duplicate tile, discarded in the epilogue
    wptr[i] = a.wp + ((size_t)(g * a.n_mt + mt) * a.n_cc) * a.K * (RTG_CK * TM) + lane;
  }
  const int n_steps = a.n_cc * a.K;
  float acur[MT][CPN], anext[MT][CPN];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int cp = 0; cp < CPN; ++cp) acur[i][cp] = wptr[i][cp * 64];

  gload(0);
  swrite(lds);
  __syncthreads();

  int step = 0;
  for (int cc = 0; cc < a.n_cc; ++cc) {
    const float* buf = lds + (cc & 1) * bufsz;
    if (cc + 1 < a.n_cc) gload(cc + 1);
    for (int tap = 0; tap < a.K; ++tap, ++step) {
      // prefetch the next step's A fragments (64 consecutive floats per fragment, L2 resident)
      if (step + 1 < n_steps) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int cp = 0; cp < CPN; ++cp) anext[i][cp] = wptr[i][(size_t)(step + 1) * (RTG_CK * TM) + cp * 64];
      }
      const int td = tap * a.dil;
      const int tapoff = (a.stride == 1) ? td : (td % a.stride) * a.PH + td / a.stride;
      const float* bp = buf + bbase + tapoff;
#pragma unroll
      for (int cp = 0; cp < CPN; ++cp) {
        float bf[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) bf[j] = bp[cp * KK * a.ROW + j * TM];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[i][j] = M::run(acur[i][cp], bf[j], acc[i][j]);
      }
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int cp = 0; cp < CPN; ++cp) acur[i][cp] = anext[i][cp];
    }
    if (cc + 1 < a.n_cc) swrite(lds + ((cc + 1) & 1) * bufsz);
    __syncthreads();
  }

  // ---- epilogue
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    if (mt0 + i >= a.n_mt) continue;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int q = q_blk + (wn * NT + j) * TM + n_lane;
      if (q >= a.Q) continue;
#pragma unroll
      for (int r = 0; r < M::NREG; ++r) {
        const int m = (mt0 + i) * TM + M::row(lane, r);
        if (m >= a.Mg) continue;
        const int mrow = g * a.Mg + m;
        int ch = mrow, u = q;
        if (a.shuf_S > 1) {
          ch = mrow / a.shuf_S;
          u = q * a.shuf_S + (mrow - ch * a.shuf_S) - a.shuf_P;
          if (u < 0 || u >= a.out_L) continue;
        }
        float* dst = a.out;
        size_t idx;
        if (a.out_split > 0) {
          if (ch >= a.out_split) {
            dst = a.out2;
            idx = ((size_t)b * (a.out_C - a.out_split) + (ch - a.out_split)) * a.out_L + u;
          } else {
            idx = ((size_t)b * a.out_split + ch) * a.out_L + u;
          }
          if (!dst) continue;
        } else {
          idx = ((size_t)b * a.out_C + ch) * a.out_L + u;
        }
        float v = acc[i][j][r];
        if (a.bias) v += a.bias[ch];
        if (a.mask) v *= (a.mask[idx] > 0.f ? 1.f : a.mask_slope);
        if (a.res) v += a.res[idx];
        v *= a.out_scale;
        if (a.act == RTG_ACT_LRELU) v = rtg_lrelu(v, a.act_slope);
        else if (a.act == RTG_ACT_TANH) v = tanhf(v);
        if (a.accumulate) v += dst[idx];
        dst[idx] = v;
      }
    }
  }
}

struct TileCfg {
  int MT, NT, WM, WN;
};

template <int TM, int MT, int NT>
int launch(const ConvArgs& a, dim3 grid, size_t lds_bytes, hipStream_t s) {
  auto k = conv1d_mfma_kernel<TM, MT, NT>;
  if (lds_bytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return -(1000 + (int)e);
  }
  hipLaunchKernelGGL(k, grid, dim3(RTG_THREADS), lds_bytes, s, a);
  return rtg_launch_status();
}

// Patch width (floats per channel) for a block covering BN output positions.
inline int patch_width(int BN, int stride, int K, int dil) { return (BN - 1) * stride + (K - 1) * dil + 1; }

TileCfg pick_tiles(int TM, int n_mt, int Q, int B, int groups, int stride, int K, int dil) {
  static const TileCfg c32[] = {{2, 2, 1, 4}, {2, 2, 2, 2}, {2, 2, 4, 1}, {1, 2, 1, 4}, {1, 2, 2, 2}, {1, 2, 4, 1},
                                {1, 4, 1, 4}, {1, 4, 2, 2}, {2, 1, 1, 4}, {2, 1, 2, 2}, {2, 1, 4, 1}, {1, 1, 1, 4},
                                {1, 1, 2, 2}, {1, 1, 4, 1}};
  static const TileCfg c16[] = {{1, 4, 1, 4}, {1, 2, 1, 4}, {1, 1, 1, 4}, {1, 4, 2, 2}, {1, 2, 2, 2}, {1, 1, 2, 2},
                                {1, 2, 4, 1}, {1, 1, 4, 1}};
  const TileCfg* cs = TM == 32 ? c32 : c16;
  const int n = TM == 32 ? (int)(sizeof(c32) / sizeof(TileCfg)) : (int)(sizeof(c16) / sizeof(TileCfg));
  TileCfg best = {0, 0, 0, 0};
  double best_score = -1.0;
  for (int i = 0; i < n; ++i) {
    const TileCfg c = cs[i];
    const int BN = c.WN * c.NT * TM;
    if (patch_width(BN, stride, K, dil) > RTG_PW_MAX) continue;
    const int mrows = c.WM * c.MT;
    const int m_blocks = rtg_ceil_div(n_mt, mrows);
    const int q_blocks = rtg_ceil_div(Q, BN);
    const double eff = ((double)n_mt / (m_blocks * mrows)) * ((double)Q / ((double)q_blocks * BN));
    const double blocks = (double)m_blocks * q_blocks * groups * B;
    const double fill = blocks >= 512.0 ? 1.0 : blocks / 512.0;
    const double reuse = (double)(c.MT * c.NT) / (c.MT + c.NT);   // MFMAs per operand fragment fetched
    const double score = eff * fill * (0.6 + 0.4 * (reuse > 1.0 ? 1.0 : reuse));
    if (score > best_score) {
      best_score = score;
      best = c;
    }
  }
  return best;
}

}  // namespace

extern "C" int rtg_conv1d_variant(const RtgConv1dDesc* d) {
  if (!d) return RTG_ENULL;
  if ((d->tile_m != 32 && d->tile_m != 16) || d->Mg < 1 || d->Q < 1 || d->B < 1 || d->groups < 1 || d->stride < 1 ||
      d->K < 1 || d->dil < 1)
    return RTG_EINVAL;
  const TileCfg c = pick_tiles(d->tile_m, rtg_ceil_div(d->Mg, d->tile_m), d->Q, d->B, d->groups, d->stride, d->K, d->dil);
  if (c.MT == 0) return RTG_ERANGE;
  return d->tile_m * 100 + c.MT * 10 + c.NT;
}

extern "C" long long rtg_packed_size(int groups, int Mg, int Cg, int K, int tile_m) {
  if (groups < 1 || Mg < 1 || Cg < 1 || K < 1 || (tile_m != 32 && tile_m != 16)) return RTG_EINVAL;
  const long long n_mt = (Mg + tile_m - 1) / tile_m, n_cc = (Cg + RTG_CK - 1) / RTG_CK;
  return (long long)groups * n_mt * n_cc * K * RTG_CK * tile_m;
}

extern "C" int rtg_conv1d(const RtgConv1dDesc* d, const float* x1, const float* x2, const float* aux, const float* wp,
                          const float* bias, const float* mask, const float* res, float* out, float* out2,
                          void* stream) {
  if (!d || !x1 || !wp) return RTG_ENULL;
  if (d->out_split == 0 ? !out : (!out && !out2)) return RTG_ENULL;
  if (d->out_split < 0 || d->out_split >= d->out_C) return RTG_EINVAL;
  if (d->out_split > 0 && (mask || res)) return RTG_EINVAL;
  if (d->B < 1 || d->C1 < 1 || d->C2 < 0 || d->L_in < 1 || d->groups < 1 || d->Cg < 1 || d->Mg < 1 || d->K < 1 ||
      d->stride < 1 || d->dil < 1 || d->Q < 1 || d->out_C < 1 || d->out_L < 1 || d->shuf_S < 1)
    return RTG_EINVAL;
  if (d->tile_m != 32 && d->tile_m != 16) return RTG_EINVAL;
  if (d->C1 + d->C2 != d->groups * d->Cg) return RTG_EINVAL;
  if (d->C2 > 0 && !x2) return RTG_ENULL;
  if ((d->pre_mode == RTG_PRE_MUL_DLRELU || d->pre_mode == RTG_PRE_MUL_DTANH) && (!aux || d->C2 != 0)) return RTG_EINVAL;
  if (d->stride > 1 && d->dil != 1) return RTG_ERANGE;
  if ((long long)d->groups * d->Mg != (long long)d->out_C * d->shuf_S) return RTG_EINVAL;
  if (d->shuf_S == 1 && d->Q > d->out_L) return RTG_EINVAL;
  if (d->B > 65535) return RTG_ERANGE;

  ConvArgs a;
  a.x1 = x1; a.x2 = x2; a.aux = (d->pre_mode >= RTG_PRE_MUL_DLRELU) ? aux : nullptr; a.wp = wp; a.bias = bias;
  a.mask = mask; a.res = res; a.out = out; a.out2 = out2; a.out_split = d->out_split;
  a.B = d->B; a.C1 = d->C1; a.C2 = d->C2; a.L_in = d->L_in; a.groups = d->groups; a.Cg = d->Cg; a.Mg = d->Mg;
  a.K = d->K; a.stride = d->stride; a.dil = d->dil; a.pad = d->pad; a.Q = d->Q; a.out_C = d->out_C; a.out_L = d->out_L;
  a.shuf_S = d->shuf_S; a.shuf_P = d->shuf_P; a.pre_mode = d->pre_mode; a.pre_slope = d->pre_slope;
  a.mask_slope = d->mask_slope; a.out_scale = d->out_scale; a.act = d->act; a.act_slope = d->act_slope;
  a.accumulate = d->accumulate;
  const int TM = d->tile_m;
  a.n_cc = rtg_ceil_div(d->Cg, RTG_CK);
  a.n_mt = rtg_ceil_div(d->Mg, TM);

  const TileCfg c = pick_tiles(TM, a.n_mt, d->Q, d->B, d->groups, d->stride, d->K, d->dil);
  if (c.MT == 0) return RTG_ERANGE;   // even the smallest block's patch exceeds RTG_PW_MAX
  a.WM = c.WM; a.WN = c.WN;
  const int BN = c.WN * c.NT * TM;
  a.PW = patch_width(BN, d->stride, d->K, d->dil);
  int row;
  if (d->stride == 1) {
    a.PH = a.PW;
    row = a.PW;
  } else {
    a.PH = rtg_ceil_div(a.PW, d->stride) | 1;   // odd phase pitch
    row = a.PH * d->stride;
  }
  a.ROW = ((row + 15) / 32) * 32 + 16;          // == 16 (mod 32), >= row
  a.m_blocks = rtg_ceil_div(a.n_mt, c.WM * c.MT);
  const long long gy = (long long)d->groups * a.m_blocks;
  if (gy > 65535) return RTG_ERANGE;
  dim3 grid(rtg_ceil_div(d->Q, BN), (unsigned)gy, d->B);
  const size_t lds_bytes = (size_t)2 * RTG_CK * a.ROW * sizeof(float);
  hipStream_t s = (hipStream_t)stream;

#define RTG_CASE(tm, mt, nt) \
  if (TM == tm && c.MT == mt && c.NT == nt) return launch<tm, mt, nt>(a, grid, lds_bytes, s);
  RTG_CASE(32, 1, 1) RTG_CASE(32, 1, 2) RTG_CASE(32, 1, 4) RTG_CASE(32, 2, 1) RTG_CASE(32, 2, 2)
  RTG_CASE(16, 1, 1) RTG_CASE(16, 1, 2) RTG_CASE(16, 1, 4)
#undef RTG_CASE
  return RTG_ERANGE;
}
