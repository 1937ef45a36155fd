// rtg_thin2d.hip — bandwidth kernels for the FIRST Conv2d of StftDiscriminator (discrminator.py:256:
// Conv2d(2, 32, (3, 3), stride (2, 1), padding (1, 1)) over the [B, 2, 1025 / 513 / 257, frames] phase / magnitude maps).
// 18 multiply-accumulates per output: the layer moves 150-300 MB per pass and computes almost nothing, but on the matrix
// path (6 virtual channels padded to a 16-channel chunk, rows of 35-137 columns) it ran at 2-12 TFLOP/s: 0.3 ms per pass.
//   forward          a thread owns two output positions (item, row, column) and all 32 output channels: its 2 x 18 inputs
//                    in registers, the 18 x 32 weights in LDS read as broadcast 16-byte fragments (one read per 8 FMAs),
//                    32 coalesced stores per position;
//   weight gradient  the four waves of a block split the output channels (8 each): a lane walks positions 64 apart with
//                    8 x 18 + 8 accumulators in registers (8 coalesced dy loads, 18 cached x loads per position), one wave
//                    reduction per block, one split partial per block in the weight bank's layout (fixed-order reduce in
//                    rtg_weightnorm_backward).
//   backward-data    (the generator step's gradient into the STFT) a thread owns a pair of input rows (2g, 2g + 1) of one
//                    column and both input channels: row 2g receives kernel row 1 of output row g, row 2g + 1 kernel rows 2
//                    and 0 of output rows g and g + 1 — two unaligned 16-byte dy loads per output channel (the three taps
//                    are consecutive columns), the 32 x 18 weights as broadcast LDS reads.
// Exposed as thin kinds 4 (forward) and 5 (backward-data) of rtg_conv1d (rtg_thin_kind) and thin kind 4 of the
// weight-gradient shape code 7.
#include "rtg_common.h"

namespace {

using rsrc_t = __amdgpu_buffer_rsrc_t;
constexpr unsigned kOob = 0x80000000u;
constexpr int kM = 32, kCr = 2, kKH = 3, kKW = 3, kNK = kCr * kKH * kKW;       // 18 products per output
constexpr int kThreads = 256;

struct T2Args {
  const float *x, *wp, *bias, *dy, *mask, *res;
  float mask_slope, out_scale;
  float *out, *part;
  long long part_stride;
  int items, H, W, Ho, h_stride, tile_m, splits;
  int Wo, w_stride;            // output columns and column stride (forward / weight gradient; round 5: the layer along the
                               // frequency axis has stride (1, 2) where the reference layout has (2, 1))
  float pre_slope, gy_scale;
  int n_pos;                   // items * Ho * Wo
  int x_bytes, y_bytes;        // bytes of x and of the [items, 32, Ho, W] tensor (out / dy)
};

__device__ __forceinline__ float t2_load(rsrc_t r, unsigned off) {
#ifdef RTG_EXP_T2_NOLOAD
  return __builtin_bit_cast(float, off);
#else
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
#endif
}

// the 18 inputs of output position p = (item, ho, w): x[item][ci][ho * h_stride - 1 + kh][w * w_stride - 1 + kw], activation applied,
// zeros outside the map (out-of-range buffer offsets); `yoff`: element offset of (item, channel 0, ho, w) in out / dy
__device__ __forceinline__ void t2_inputs_at(const T2Args& a, rsrc_t rx, bool live, int b, int ho, int w, float (&xv)[kNK],
                                             unsigned& yoff) {
  const int hw = a.Ho * a.Wo;
  yoff = live ? (unsigned)(b * kM * hw + ho * a.Wo + w) * 4u : kOob;
#pragma unroll
  for (int ci = 0; ci < kCr; ++ci)
#pragma unroll
    for (int kh = 0; kh < kKH; ++kh) {
      const int row = ho * a.h_stride - 1 + kh;
      const bool rok = live && (unsigned)row < (unsigned)a.H;
      const unsigned rb = (unsigned)(((b * kCr + ci) * a.H + row) * a.W);
#pragma unroll
      for (int kw = 0; kw < kKW; ++kw) {
        const int col = w * a.w_stride - 1 + kw;
        const float v = t2_load(rx, (rok && (unsigned)col < (unsigned)a.W) ? (rb + (unsigned)col) * 4u : kOob);
        xv[(ci * kKH + kh) * kKW + kw] = v > 0.f ? v : v * a.pre_slope;
      }
    }
}

__device__ __forceinline__ void t2_inputs(const T2Args& a, rsrc_t rx, int p, float (&xv)[kNK], unsigned& yoff) {
  const int hw = a.Ho * a.Wo;
  const int b = p / hw, r = p - b * hw;
  const int ho = r / a.Wo, w = r - ho * a.Wo;
  t2_inputs_at(a, rx, p < a.n_pos, b, ho, w, xv, yoff);
}

__global__ __launch_bounds__(kThreads) void cin2_fwd_kernel(const T2Args a) {
  __shared__ __attribute__((aligned(16))) float wl[kNK * kM];         // [product k = (ci, kh, kw)][output channel]
  __shared__ __attribute__((aligned(16))) float bl[kM];
  // weights from the packed forward image [row tile][chunk 0][tap = kw][channel pair][kk][row], channel = ci * 3 + kh
  const int KK = 64 / a.tile_m;
  for (int e = threadIdx.x; e < kNK * kM; e += kThreads) {
    const int k = e / kM, m = e - k * kM;
    const int c = k / kKW, tap = k - c * kKW;
    const int mt = m / a.tile_m, mm = m - mt * a.tile_m;
    wl[e] = a.wp[(mt * kKW + tap) * (RTG_CK * a.tile_m) + (c / KK) * 64 + (c % KK) * a.tile_m + mm];
  }
  if (threadIdx.x < kM) bl[threadIdx.x] = a.bias ? a.bias[threadIdx.x] : 0.f;
  const rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
  const rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, a.y_bytes, 0x00020000);
  float xa[kNK], xb[kNK];
  unsigned oa, ob;
  const int p0 = (int)blockIdx.x * (2 * kThreads) + (int)threadIdx.x;
  t2_inputs(a, rx, p0, xa, oa);
  t2_inputs(a, rx, p0 + kThreads, xb, ob);
  __syncthreads();
  const unsigned chb = (unsigned)(a.Ho * a.Wo) * 4u;                    // bytes per output channel plane
#pragma unroll
  for (int g = 0; g < kM / 4; ++g) {
    float va[4], vb[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) va[r] = vb[r] = bl[4 * g + r];
#ifdef RTG_EXP_T2_NOFMA
    if (g > 0) continue;
#endif
#pragma unroll
    for (int k = 0; k < kNK; ++k) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float wv = wl[k * kM + 4 * g + r];                      // (every lane the same address: broadcast reads)
        va[r] = __builtin_fmaf(xa[k], wv, va[r]);
        vb[r] = __builtin_fmaf(xb[k], wv, vb[r]);
      }
    }
#ifdef RTG_EXP_T2_NOSTORE
    if (va[0] + va[1] + va[2] + va[3] + vb[0] + vb[1] + vb[2] + vb[3] == 12345.678f)
#endif
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, va[r]), ro, oa, (unsigned)(4 * g + r) * chb, 0);
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, vb[r]), ro, ob, (unsigned)(4 * g + r) * chb, 0);
    }
  }
}

__global__ __launch_bounds__(kThreads) void cin2_wgrad_kernel(const T2Args a) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);    // output channels 8 * wave .. + 7
  const rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
  const rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, a.y_bytes, 0x00020000);
  const int per = (a.n_pos + a.splits - 1) / a.splits;
  const int p_lo = (int)blockIdx.x * per;
  const int p_hi = p_lo + per < a.n_pos ? p_lo + per : a.n_pos;
  const unsigned chb = (unsigned)(a.Ho * a.Wo) * 4u;
  float acc[8][kNK], bacc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    bacc[j] = 0.f;
#pragma unroll
    for (int k = 0; k < kNK; ++k) acc[j][k] = 0.f;
  }
  // (item, row, column) of this lane's position, advanced 64 positions per iteration without divisions (W >= 5: a few
  // conditional subtractions)
  int pb, pho, pw;
  {
    const int p = p_lo + lane, hw = a.Ho * a.Wo;
    pb = p / hw;
    const int r = p - pb * hw;
    pho = r / a.Wo;
    pw = r - pho * a.Wo;
  }
  const int wq = 64 / a.Wo, wr = 64 - wq * a.Wo;                         // 64 positions = wq rows + wr columns
  for (int p = p_lo + lane; p < p_hi; p += 64) {
    float xv[kNK], gy[8];
    unsigned yo;
    t2_inputs_at(a, rx, true, pb, pho, pw, xv, yo);
    pw += wr;
    pho += wq + (pw >= a.Wo ? 1 : 0);
    pw -= pw >= a.Wo ? a.Wo : 0;
    while (pho >= a.Ho) {                                               // (one item at most, except on maps of a few rows)
      pho -= a.Ho;
      ++pb;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
      gy[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rd, yo, (unsigned)(8 * wave + j) * chb, 0));
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      bacc[j] += gy[j];
#pragma unroll
      for (int k = 0; k < kNK; ++k) acc[j][k] = __builtin_fmaf(gy[j], xv[k], acc[j][k]);
    }
  }
  // this block's partial: [row m][column (ci * 3 + kh) * 3 + kw], then the bias column
  float* wpart = a.part + (size_t)blockIdx.x * a.part_stride;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int m = 8 * wave + j;
#pragma unroll
    for (int k = 0; k < kNK; ++k) {
      const float s = rtg_wave_sum(acc[j][k]);
      if (lane == 0) wpart[m * kNK + k] = s * a.gy_scale;
    }
    const float sb = rtg_wave_sum(bacc[j]);
    if (lane == 0) wpart[kM * kNK + m] = sb * a.gy_scale;
  }
}

// backward-data: thread = (item, row pair g, column w); dy is [items, 32, Ho, W], the result [items, 2, H, W]
__global__ __launch_bounds__(kThreads) void cin2_dgrad_kernel(const T2Args a) {
  __shared__ __attribute__((aligned(16))) float wl[kM * 20];          // [co][kernel row][tap][ci], padded to 20 floats
  // weights from the packed backward image [row tile 0][chunk][tap][channel quad][kk][row]: rows = ci, channels = (kh, co),
  // taps along the columns already flipped (tap t multiplies dy column w - 1 + t)
  const int KK = 64 / a.tile_m;
  for (int e = threadIdx.x; e < kM * 18; e += kThreads) {
    const int co = e / 18, r = e - co * 18;
    const int kh = r / 6, tap = (r - kh * 6) >> 1, ci = r & 1;
    const int c = kh * kM + co, cc = c / RTG_CK, c16 = c - cc * RTG_CK;
    wl[co * 20 + r] = a.wp[(cc * kKW + tap) * (RTG_CK * a.tile_m) + (c16 / KK) * 64 + (c16 % KK) * a.tile_m + ci];
  }
  const rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, a.y_bytes, 0x00020000);
  const int G = (a.H + 1) >> 1;
  const int n_thr = a.items * G * a.W;
  const int p = (int)blockIdx.x * kThreads + (int)threadIdx.x;
  const bool live = p < n_thr;
  const int gw = G * a.W;
  const int b = p / gw, r = p - b * gw;
  const int g = r / a.W, w = r - g * a.W;
  // byte offsets of dy[b][0][g][w - 1] and dy[b][0][g + 1][w - 1]; output rows past the map load from out of range
  const int hw = a.Ho * a.W;
  const int e0 = b * kM * hw + g * a.W + w - 1;
  const bool r0 = live && g < a.Ho, r1 = live && g + 1 < a.Ho;
  const bool lok = w > 0, rok = w + 1 < a.W;                            // taps 0 / 2 inside the row
  __syncthreads();
  float a00 = 0.f, a01 = 0.f, a10 = 0.f, a11 = 0.f;                    // [row 2g / 2g + 1][ci]
#pragma unroll 4
  for (int co = 0; co < kM; ++co) {
    float u[3], v[3];                                                  // dy rows g and g + 1, columns w - 1 .. w + 1
    const int eo = e0 + co * hw;
    if (co == 0) {
      // (the 16 bytes of the very first element would start before the buffer: three 4-byte loads; element -1 wraps to an
      // offset past the end and reads zero)
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        u[t] = t2_load(rd, r0 ? (unsigned)(eo + t) * 4u : kOob);
        v[t] = t2_load(rd, r1 ? (unsigned)(eo + a.W + t) * 4u : kOob);
      }
    } else {
      const f32x4 lu = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rd, r0 ? (unsigned)eo * 4u : kOob, 0, 0));
      const f32x4 lv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rd, r1 ? (unsigned)(eo + a.W) * 4u : kOob, 0, 0));
#pragma unroll
      for (int t = 0; t < 3; ++t) { u[t] = lu[t]; v[t] = lv[t]; }
    }
    u[0] = lok ? u[0] : 0.f; v[0] = lok ? v[0] : 0.f;
    u[2] = rok ? u[2] : 0.f; v[2] = rok ? v[2] : 0.f;
    const float* wc = wl + co * 20;                                    // [kh][tap][ci]
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      a00 = __builtin_fmaf(wc[6 + 2 * t], u[t], a00);                  // kernel row 1, output row g -> input row 2g
      a01 = __builtin_fmaf(wc[6 + 2 * t + 1], u[t], a01);
      a10 = __builtin_fmaf(wc[12 + 2 * t], u[t], a10);                 // kernel row 2, output row g -> input row 2g + 1
      a11 = __builtin_fmaf(wc[12 + 2 * t + 1], u[t], a11);
      a10 = __builtin_fmaf(wc[2 * t], v[t], a10);                      // kernel row 0, output row g + 1 -> input row 2g + 1
      a11 = __builtin_fmaf(wc[2 * t + 1], v[t], a11);
    }
  }
  if (!live) return;
  const float res[2][2] = {{a00, a01}, {a10, a11}};
#pragma unroll
  for (int rr = 0; rr < 2; ++rr) {
    const int h = 2 * g + rr;
    if (h >= a.H) continue;
#pragma unroll
    for (int ci = 0; ci < kCr; ++ci) {
      const size_t o = ((size_t)(b * kCr + ci) * a.H + h) * a.W + w;
      float val = res[rr][ci];
      if (a.mask) val *= a.mask[o] > 0.f ? 1.f : a.mask_slope;
      if (a.res) val += a.res[o];
      a.out[o] = val * a.out_scale;
    }
  }
}

// backward-data of the layer run along the frequency axis (WNConv wt: stride (1, 2) on the [items, 2, frames, F] map): thread =
// (item, row r, column pair g); column 2g receives kernel column 1 of output column g, column 2g + 1 kernel columns 2 and 0 of
// output columns g and g + 1, from the output rows r + 1 - kr of the three kernel rows — six dy values per output channel as
// three 8-byte loads, consecutive threads consecutive pairs.  dy is [items, 32, H, Wo], the result [items, 2, H, W].
__global__ __launch_bounds__(kThreads) void cin2_dgrad_cols_kernel(const T2Args a) {
  __shared__ __attribute__((aligned(16))) float wl[kM * 20];          // [co][kernel row][kernel column][ci], padded to 20 floats
  // weights from the packed polyphase backward image [row tile 0][chunk][tap 2][channel quad][kk][row]: rows = (ci, phase),
  // channels = (kernel row, co); kernel column jj = phase + (1 - tap) * 2
  const int KK = 64 / a.tile_m;
  for (int e = threadIdx.x; e < kM * 18; e += kThreads) {
    const int co = e / 18, r = e - co * 18;
    const int kr = r / 6, kc = (r - kr * 6) >> 1, ci = r & 1;
    const int ph = kc & 1, tap = 1 - (kc >> 1);
    const int c = kr * kM + co, cc = c / RTG_CK, c16 = c - cc * RTG_CK;
    wl[co * 20 + r] = a.wp[(cc * 2 + tap) * (RTG_CK * a.tile_m) + (c16 / KK) * 64 + (c16 % KK) * a.tile_m + ci * 2 + ph];
  }
  const rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, a.y_bytes, 0x00020000);
  const int G = (a.W + 1) >> 1;
  const int n_thr = a.items * a.H * G;
  const int p = (int)blockIdx.x * kThreads + (int)threadIdx.x;
  const bool live = p < n_thr;
  const int hg = a.H * G;
  const int b = p / hg, rem = p - b * hg;
  const int r = rem / G, g = rem - r * G;
  const int hw = a.H * a.Wo;                                            // (output rows = input rows: row stride 1, "same")
  const bool c0 = live && g < a.Wo, c1 = live && g + 1 < a.Wo;
  __syncthreads();
  float a00 = 0.f, a01 = 0.f, a10 = 0.f, a11 = 0.f;                    // [column 2g / 2g + 1][ci]
#pragma unroll 4
  for (int co = 0; co < kM; ++co) {
    const float* wc = wl + co * 20;                                    // [kr][kc][ci]
#pragma unroll
    for (int kr = 0; kr < 3; ++kr) {
      const int ro = r + 1 - kr;
      const bool rok = (unsigned)ro < (unsigned)a.H;
      const unsigned e = (unsigned)((b * kM + co) * hw + ro * a.Wo + g);
      const float u0 = t2_load(rd, (rok && c0) ? e * 4u : kOob), u1 = t2_load(rd, (rok && c1) ? (e + 1u) * 4u : kOob);
      a00 = __builtin_fmaf(wc[kr * 6 + 2], u0, a00);                   // kernel column 1, output column g -> column 2g
      a01 = __builtin_fmaf(wc[kr * 6 + 3], u0, a01);
      a10 = __builtin_fmaf(wc[kr * 6 + 4], u0, a10);                   // kernel column 2, output column g -> column 2g + 1
      a11 = __builtin_fmaf(wc[kr * 6 + 5], u0, a11);
      a10 = __builtin_fmaf(wc[kr * 6 + 0], u1, a10);                   // kernel column 0, output column g + 1 -> column 2g + 1
      a11 = __builtin_fmaf(wc[kr * 6 + 1], u1, a11);
    }
  }
  if (!live) return;
  const float res[2][2] = {{a00, a01}, {a10, a11}};
#pragma unroll
  for (int cc = 0; cc < 2; ++cc) {
    const int w = 2 * g + cc;
    if (w >= a.W) continue;
#pragma unroll
    for (int ci = 0; ci < kCr; ++ci) {
      const size_t o = ((size_t)(b * kCr + ci) * a.H + r) * a.W + w;
      float val = res[cc][ci];
      if (a.mask) val *= a.mask[o] > 0.f ? 1.f : a.mask_slope;
      if (a.res) val += a.res[o];
      a.out[o] = val * a.out_scale;
    }
  }
}

template <class D>
bool t2_shape_ok(const D* d) {
  if (d->groups != 1 || d->C2 != 0 || d->h_k != kKH || d->K != kKW || d->C1 != kCr * kKH || d->Cg != d->C1 || d->Mg != kM) return false;
  if (d->stride < 1 || d->stride > 2 || d->dil != 1 || d->pad != 1 || d->h_pad != 1 || d->h_stride < 1 || d->h_stride > 2) return false;
  if (d->h_in < 1 || d->h_n < 1 || d->B % d->h_n != 0 || d->Q != (d->L_in + 2 - kKW) / d->stride + 1) return false;
  if (d->h_n != (d->h_in + 2 - kKH) / d->h_stride + 1) return false;
  if (d->pre_mode != RTG_PRE_NONE && d->pre_mode != RTG_PRE_LRELU) return false;
  const long long items = d->B / d->h_n;
  if (items * kCr * d->h_in * d->L_in * 4 >= (1ll << 31) || items * kM * d->h_n * (long long)d->Q * 4 >= (1ll << 31)) return false;
  return true;
}

template <class D>
void t2_fill(const D* d, T2Args* a) {
  a->items = d->B / d->h_n; a->H = d->h_in; a->W = d->L_in; a->Ho = d->h_n; a->h_stride = d->h_stride;
  a->Wo = d->Q; a->w_stride = d->stride;
  a->pre_slope = d->pre_mode == RTG_PRE_LRELU ? d->pre_slope : 1.f;
  a->n_pos = a->items * a->Ho * a->Wo;
  a->x_bytes = a->items * kCr * a->H * a->W * 4;
  a->y_bytes = a->items * kM * a->Ho * a->Wo * 4;
}

}  // namespace

bool rtg_thin2d_fwd_ok(const RtgConv1dDesc* d) {
  if (!t2_shape_ok(d) || d->h_mode != 0 || d->bf16 || d->tap_major || d->shuf_S != 1 || d->out_split != 0 || d->accumulate) return false;
  if (d->out_C != kM || d->out_L != d->Q || d->act != RTG_ACT_NONE || d->out_scale != 1.f) return false;
  if (d->tile_m != 32 && d->tile_m != 16) return false;
  return true;
}

int rtg_thin2d_fwd_launch(const RtgConv1dDesc* d, const float* x, const float* wp, const float* bias, const float* mask,
                          const float* res, float* out, hipStream_t s) {
  if (!rtg_thin2d_fwd_ok(d) || mask || res) return RTG_EINVAL;
  if (!x || !wp || !out) return RTG_ENULL;
  T2Args a = {};
  t2_fill(d, &a);
  a.x = x; a.wp = wp; a.bias = bias; a.out = out; a.tile_m = d->tile_m;
  RTG_KLAUNCH(cin2_fwd_kernel, dim3((unsigned)rtg_ceil_div(a.n_pos, 2 * kThreads)), dim3(kThreads), 0, s, a);
  return rtg_launch_status();
}

// ... of the layer along the frequency axis: the 2-tap polyphase operator of its column stride 2 (rows = (ci, phase), shuffle
// store), row stride 1
static bool t2_dgrad_cols_ok(const RtgConv1dDesc* d) {
  if (d->h_mode != 1 || d->groups != 1 || d->C2 != 0 || d->Mg != kCr * 2 || d->C1 != kM * kKH || d->Cg != d->C1) return false;
  if (d->h_k != kKH || d->K != 2 || d->stride != 1 || d->dil != 1 || d->pad != 1 || d->shuf_S != 2 || d->shuf_P != 1) return false;
  if (d->h_pad != 1 || d->h_stride != 1 || d->h_in < 1 || d->h_n != d->h_in || d->B % d->h_n != 0 || d->out_C != kCr) return false;
  if (d->out_L < 1 || d->L_in != (d->out_L + 2 - kKW) / 2 + 1) return false;
  if (d->bf16 || d->io_bf16 || d->tap_major || d->out_split != 0 || d->accumulate || d->act != RTG_ACT_NONE) return false;
  if (d->pre_mode != RTG_PRE_NONE || (d->tile_m != 16 && d->tile_m != 32)) return false;
  const long long items = d->B / d->h_n;
  if (items * kM * d->h_in * d->L_in * 4 >= (1ll << 31) || items * kCr * d->h_n * (long long)d->out_L * 4 >= (1ll << 31)) return false;
  return true;
}

// backward-data descriptor of the same layer (Conv2dFn.backward): clips = (item, input row), channels = (kernel row, co)
bool rtg_thin2d_dgrad_ok(const RtgConv1dDesc* d) {
  if (t2_dgrad_cols_ok(d)) return true;
  if (d->h_mode != 1 || d->groups != 1 || d->C2 != 0 || d->Mg != kCr || d->C1 != kM * kKH || d->Cg != d->C1) return false;
  if (d->h_k != kKH || d->K != kKW || d->stride != 1 || d->dil != 1 || d->pad != 1 || d->h_pad != 1 || d->h_stride != 2) return false;
  if (d->h_in < 1 || d->h_n < 1 || d->B % d->h_n != 0 || d->Q != d->L_in || d->out_L != d->Q || d->out_C != kCr) return false;
  if (d->h_in != (d->h_n + 2 - kKH) / 2 + 1) return false;              // (h_in = output rows of the layer, h_n = its input rows)
  if (d->bf16 || d->tap_major || d->shuf_S != 1 || d->out_split != 0 || d->accumulate || d->act != RTG_ACT_NONE) return false;
  if (d->pre_mode != RTG_PRE_NONE || (d->tile_m != 16 && d->tile_m != 32)) return false;
  const long long items = d->B / d->h_n;
  if (items * kM * d->h_in * d->L_in * 4 >= (1ll << 31) || items * kCr * d->h_n * d->L_in * 4 >= (1ll << 31)) return false;
  return true;
}

int rtg_thin2d_dgrad_launch(const RtgConv1dDesc* d, const float* dy, const float* wp, const float* mask, const float* res,
                            float* out, hipStream_t s) {
  if (!rtg_thin2d_dgrad_ok(d)) return RTG_EINVAL;
  if (!dy || !wp || !out) return RTG_ENULL;
  if (t2_dgrad_cols_ok(d)) {
    T2Args c = {};
    c.items = d->B / d->h_n; c.H = d->h_n; c.W = d->out_L; c.Wo = d->L_in; c.Ho = d->h_in; c.tile_m = d->tile_m;
    c.dy = dy; c.wp = wp; c.mask = mask; c.res = res; c.out = out;
    c.mask_slope = d->mask_slope; c.out_scale = d->out_scale;
    c.y_bytes = c.items * kM * c.H * c.Wo * 4;
    const long long n = (long long)c.items * c.H * ((c.W + 1) / 2);
    RTG_KLAUNCH(cin2_dgrad_cols_kernel, dim3((unsigned)rtg_ceil_div(n, kThreads)), dim3(kThreads), 0, s, c);
    return rtg_launch_status();
  }
  T2Args a = {};
  a.items = d->B / d->h_n; a.H = d->h_n; a.W = d->L_in; a.Ho = d->h_in; a.h_stride = 2; a.tile_m = d->tile_m;
  a.dy = dy; a.wp = wp; a.mask = mask; a.res = res; a.out = out;
  a.mask_slope = d->mask_slope; a.out_scale = d->out_scale;
  a.y_bytes = a.items * kM * a.Ho * a.W * 4;
  const long long n_thr = (long long)a.items * ((a.H + 1) / 2) * a.W;
  RTG_KLAUNCH(cin2_dgrad_kernel, dim3((unsigned)rtg_ceil_div(n_thr, kThreads)), dim3(kThreads), 0, s, a);
  return rtg_launch_status();
}

bool rtg_thin2d_wgrad_ok(const RtgWgradDesc* d) {
  if (!t2_shape_ok(d) || d->gy_mode != RTG_PRE_NONE || d->dy_L != d->Q) return false;
  return true;
}

// one block per split; a lane should see a few dozen positions (the wave reduction at the end is ~150 values)
int rtg_thin2d_wgrad_splits(const RtgWgradDesc* d) {
  if (!rtg_thin2d_wgrad_ok(d)) return RTG_EINVAL;
  const long long n_pos = (long long)d->B * d->Q;
  long long s = n_pos / (64 * 48);
  if (s > 512) s = 512;
  return (int)(s < 1 ? 1 : s);
}

int rtg_thin2d_wgrad_launch(const RtgWgradDesc* d, const float* x, const float* dy, float* part, hipStream_t s) {
  if (!rtg_thin2d_wgrad_ok(d) || d->splits != rtg_thin2d_wgrad_splits(d)) return RTG_EINVAL;
  if (!x || !dy || !part) return RTG_ENULL;
  T2Args a = {};
  t2_fill(d, &a);
  a.x = x; a.dy = dy; a.part = part; a.part_stride = d->part_stride; a.splits = d->splits; a.gy_scale = d->gy_scale;
  RTG_KLAUNCH(cin2_wgrad_kernel, dim3((unsigned)d->splits), dim3(kThreads), 0, s, a);
  return rtg_launch_status();
}
