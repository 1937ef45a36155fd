// rtg_stft.hip — framed rFFT of get_stft_torch (retunegan/audio.py:150-170) with the |.|, log, angle/PI and mel
// epilogues of multi_stft_loss (retunegan/models/loss.py:32-52) fused, and its backward.
//
// Round 5.  A frame is n_fft REAL samples (only the win = n_fft/2 in the middle are non-zero), so its spectrum comes from a
// complex FFT of HALF the length: z[m] = x[2m] + i x[2m+1], Z = FFT_M(z) with M = n_fft/2, then
//     X[k] = (Z[k] + conj Z[M-k]) / 2  +  W_N^k (Z[k] - conj Z[M-k]) / (2i),   k = 0 .. M.
// FFT_M is a radix-4 Stockham autosort in LDS (one radix-2 pass where M is not a power of 4): 5 passes for n_fft 2048 where
// the radix-2 transform of n_fft complex points took 11, each with a barrier and an LDS round trip — what the kernel's time
// was made of.  M/4 threads own a frame (one radix-4 butterfly per thread and pass), a 256-thread block takes 1 / 2 / 4
// frames (n_fft 2048 / 1024 / 512).  Twiddles from the fp64-rounded table (L2 resident).  The epilogue runs on the M + 1
// bins straight from LDS; DC and Nyquist come out exactly real, like an r2c FFT's.  HBM traffic per frame = win floats in,
// (mel | spec | re, im) out; with RtgStftDesc.spec_T the spectrum stores are coalesced.
// Backward = the adjoint of a real-input DFT: y[n] = Re sum_f conj(G_f) W^(fn) is the DFT of the Hermitian extension of
// c_f conj(G_f) (c = 1/2 inside, 1 at DC / Nyquist) and real, so it is one FFT_M too: Z'[k] = (H[k] + H[k+M]) +
// i (H[k] - H[k+M]) W_N^k, FFT_M(Z')[m] = y[2m] + i y[2m+1]; windowed into a per-frame workspace, then a gather-form
// overlap-add that also folds the reflect padding back.
#include "rtg_common.h"

namespace {

#define RTG_PI_REF 3.14159265358979f   // retunegan/utils.py:12

struct cpx {
  float x, y;
};

// exp(-2 pi i j / N) for 0 <= j < N from the table of N/2 (cos, sin) pairs of the angles 2 pi k / N
__device__ __forceinline__ cpx tw_at(const float* __restrict__ tw, int N, int j) {
  const int h = N >> 1;
  const bool up = j >= h;
  const int k = up ? j - h : j;
  const float c = tw[k], s = tw[h + k];
  return up ? cpx{-c, s} : cpx{c, -s};
}
__device__ __forceinline__ cpx cmul(cpx a, cpx b) { return cpx{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }

// FFT of M complex points in LDS (radix-4 Stockham autosort, decimation in frequency, forward sign), by the TPF = M/4
// threads t = 0 .. TPF-1 of one frame; every pass ends in a block barrier (all frames of the block walk the same passes).
// N = 2M: the twiddle table's length.  Returns the buffer that holds the result.
__device__ __forceinline__ cpx* fft_half(cpx* a, cpx* b, int M, int t0, int TPF, const float* __restrict__ tw) {
  const int N = 2 * M, q4 = M >> 2;
  cpx *src = a, *dst = b;
  int s = 1;
  for (; 4 * s <= M; s <<= 2) {
    // butterfly idx = q + s p (q < s): inputs idx + j M/4, outputs q + s (4p + j) = idx + 3 s p + s j, twiddles
    // exp(-2 pi i j (s p) / M) = table[2 j s p]
   for (int t = t0; t < q4; t += TPF) {                       // (one butterfly per thread up to n_fft 2048, two at 4096)
    const int k = t & ~(s - 1);
    const cpx x0 = src[t], x1 = src[t + q4], x2 = src[t + 2 * q4], x3 = src[t + 3 * q4];
    const cpx apc{x0.x + x2.x, x0.y + x2.y}, amc{x0.x - x2.x, x0.y - x2.y};
    const cpx bpd{x1.x + x3.x, x1.y + x3.y}, bmd{x1.x - x3.x, x1.y - x3.y};
    const cpx y0{apc.x + bpd.x, apc.y + bpd.y};
    const cpx y1{amc.x + bmd.y, amc.y - bmd.x};               // amc - i (b - d)
    const cpx y2{apc.x - bpd.x, apc.y - bpd.y};
    const cpx y3{amc.x - bmd.y, amc.y + bmd.x};               // amc + i (b - d)
    cpx* o = dst + t + 3 * k;
    o[0] = y0;
    if (k == 0) {
      o[s] = y1; o[2 * s] = y2; o[3 * s] = y3;
    } else {
      o[s] = cmul(y1, tw_at(tw, N, 2 * k));
      o[2 * s] = cmul(y2, tw_at(tw, N, 4 * k));
      o[3 * s] = cmul(y3, tw_at(tw, N, 6 * k));
    }
   }
    __syncthreads();
    cpx* tmp = src; src = dst; dst = tmp;
  }
  if (s < M) {                                                // M = 2 * 4^n: one radix-2 pass, s = M/2 (twiddle 1)
    const int half = M >> 1;
    for (int t = t0; t < q4; t += TPF)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int idx = t + u * q4;                             // M/2 butterflies, two per thread and round
      const int k = idx & ~(s - 1);                           // (= 0: idx < s)
      const cpx p = src[idx], r = src[idx + half];
      dst[idx + k] = cpx{p.x + r.x, p.y + r.y};
      dst[idx + k + s] = cpx{p.x - r.x, p.y - r.y};
    }
    __syncthreads();
    cpx* tmp = src; src = dst; dst = tmp;
  }
  return src;
}

constexpr int kStftThreads = 256;

__device__ __forceinline__ void stft_fwd_body(const RtgStftDesc& d, const float* __restrict__ y,
                                              const float* __restrict__ window, const float* __restrict__ twiddle,
                                              const int* __restrict__ mel_lo, const int* __restrict__ mel_len,
                                              const int* __restrict__ mel_woff, const float* __restrict__ mel_w,
                                              float* mel, float* spec, float* re_out, float* im_out, int fblock, int b) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int N = d.n_fft, M = N >> 1, F = M + 1;
  const int TPF = min(M >> 2, kStftThreads), FPB = kStftThreads / TPF;   // threads per frame, frames per block
  const int fl = threadIdx.x / TPF, t = threadIdx.x - fl * TPF;
  const int frame = fblock * FPB + fl;
  const bool live = frame < d.frames;
  float* base = smem + (size_t)fl * (4 * M + F + 3);         // per frame: two buffers of M complex points, F magnitudes
  cpx* A = reinterpret_cast<cpx*>(base);
  cpx* Bf = A + M;
  float* S = reinterpret_cast<float*>(Bf + M);
  const int lpad = (N - d.win) / 2;
  const float* yb = y + (size_t)b * d.T;

  for (int m = t; m < M; m += TPF) {
    float v[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int i = 2 * m + h, n = i - lpad;
      v[h] = 0.f;
      if (live && n >= 0 && n < d.win) {
        int tt = frame * d.hop + i - N / 2;                  // centre=True: padded index - n_fft/2
        if (tt < 0) tt = -tt;
        if (tt >= d.T) tt = 2 * (d.T - 1) - tt;
        v[h] = yb[tt] * window[n];
      }
    }
    A[m] = cpx{v[0], v[1]};
  }
  __syncthreads();
  const cpx* Z = fft_half(A, Bf, M, t, TPF, twiddle);

  const size_t fo = ((size_t)b * d.frames + frame) * F;     // [B][frames][F] scratch layout for the backward
  for (int f = t; f < F; f += TPF) {
    // X[f] = (Z[f] + conj Z[M-f]) / 2 + W^f (Z[f] - conj Z[M-f]) / (2i); DC and Nyquist (f = 0, M) are exactly real:
    // a real-to-complex FFT (torch.stft / pocketfft) returns imag = +0.0 there, so angle() is exactly 0 or +pi
    float re, im;
    if (f == 0 || f == M) {
      re = f == 0 ? Z[0].x + Z[0].y : Z[0].x - Z[0].y;
      im = 0.f;
    } else {
      const cpx p = Z[f], r = Z[M - f];
      const float ex = 0.5f * (p.x + r.x), ey = 0.5f * (p.y - r.y);        // even part
      const float ox = 0.5f * (p.y + r.y), oy = -0.5f * (p.x - r.x);       // odd part (Z[f] - conj Z[M-f]) / (2i)
      const cpx w = tw_at(twiddle, N, f);
      re = ex + w.x * ox - w.y * oy;
      im = ey + w.x * oy + w.y * ox;
    }
    const float rr = re + 1e-9f;
    const float mag = sqrtf(rr * rr + im * im);
    S[f] = mag;
    if (!live) continue;
    if (re_out) {
      re_out[fo + f] = re;
      im_out[fo + f] = im;
    }
    if (spec) {
      // [B][2][F][frames] (the reference's stack, loss.py:36-44: lanes 4 bytes x `frames` apart) or, spec_T, [B][2][frames][F]:
      // a frame's bins are consecutive — coalesced stores, and the layout the spectrogram discriminators walk
      const size_t so = d.spec_T ? (((size_t)b * 2) * d.frames + frame) * F + f : (((size_t)b * 2) * F + f) * d.frames + frame;
      spec[so] = logf(mag);
      spec[so + (size_t)F * d.frames] = atan2f(im, re) / RTG_PI_REF;
    }
  }
  __syncthreads();
  if (mel && live) {
    for (int m = t; m < d.n_mel; m += TPF) {
      const int lo = mel_lo[m], len = mel_len[m];
      const float* w = mel_w + mel_woff[m];
      float acc = 0.f;
      for (int i = 0; i < len; ++i) acc += w[i] * S[lo + i];
      mel[((size_t)b * d.n_mel + m) * d.frames + frame] = acc;
    }
  }
}

// Several (resolution, signal) jobs in ONE launch (round 6): multi_stft_loss runs three resolutions on the real and the
// generated wave — six forward launches of 14-17 us that move 6 MB each (launch-shaped, not byte-shaped), three backward
// frame launches and three overlap-adds on the generator's serial chain.  grid.x enumerates the jobs' frame blocks one job
// after the other (blk_end = running sums), grid.y the clips; a block whose clip index is past its job's batch returns.
constexpr int kStftMaxJobs = RTG_STFT_MAX_JOBS;
using StftFwdJob = RtgStftFwdJob;          // (include/rtg.h: descriptor + the operands of rtg_stft_forward)
struct StftFwdJobs {
  int n;
  int blk_end[kStftMaxJobs];
  StftFwdJob j[kStftMaxJobs];
};

__global__ __launch_bounds__(kStftThreads) void stft_fwd_kernel(RtgStftDesc d, const float* __restrict__ y,
                                                                const float* __restrict__ window,
                                                                const float* __restrict__ twiddle,
                                                                const int* __restrict__ mel_lo,
                                                                const int* __restrict__ mel_len,
                                                                const int* __restrict__ mel_woff,
                                                                const float* __restrict__ mel_w, float* mel, float* spec,
                                                                float* re_out, float* im_out) {
  stft_fwd_body(d, y, window, twiddle, mel_lo, mel_len, mel_woff, mel_w, mel, spec, re_out, im_out, blockIdx.x, blockIdx.y);
}

__global__ __launch_bounds__(kStftThreads) void stft_fwd_multi_kernel(const StftFwdJobs js) {
  int ji = 0;
  while (ji + 1 < js.n && (int)blockIdx.x >= js.blk_end[ji]) ++ji;           // (block-uniform)
  const StftFwdJob& q = js.j[ji];
  if ((int)blockIdx.y >= q.d.B) return;
  stft_fwd_body(q.d, q.y, q.window, q.twiddle, q.mel_lo, q.mel_len, q.mel_woff, q.mel_w, q.mel, q.spec, q.re, q.im,
                (int)blockIdx.x - (ji ? js.blk_end[ji - 1] : 0), blockIdx.y);
}

__device__ __forceinline__ void stft_bwd_frame_body(const RtgStftDesc& d, const float* __restrict__ re_in,
                                                    const float* __restrict__ im_in, const float* __restrict__ dmel,
                                                    const float* __restrict__ dspec, const float* __restrict__ window,
                                                    const float* __restrict__ twiddle, const int* __restrict__ binmel_idx,
                                                    const float* __restrict__ binmel_w, float* __restrict__ frame_ws,
                                                    int fblock, int b) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int N = d.n_fft, M = N >> 1, F = M + 1;
  const int TPF = min(M >> 2, kStftThreads), FPB = kStftThreads / TPF;
  const int fl = threadIdx.x / TPF, t = threadIdx.x - fl * TPF;
  const int frame = fblock * FPB + fl;
  const bool live = frame < d.frames;
  // per frame: two buffers of M complex points, the F cotangents conj-weighted H[f] (2 F floats), n_mel mel cotangents
  float* base = smem + (size_t)fl * (4 * M + 2 * F + 256 + 2);
  cpx* A = reinterpret_cast<cpx*>(base);
  cpx* Bf = A + M;
  cpx* H = Bf + M;                                           // H[f] = c_f conj(G_f), f = 0 .. M
  float* dm = reinterpret_cast<float*>(H + F);               // n_mel cotangents of this frame
  const int lpad = (N - d.win) / 2;

  if (dmel)
    for (int m = t; m < d.n_mel; m += TPF) dm[m] = live ? dmel[((size_t)b * d.n_mel + m) * d.frames + frame] : 0.f;
  __syncthreads();
  const size_t fo = ((size_t)b * d.frames + frame) * F;
  for (int f = t; f < F; f += TPF) {
    float gr = 0.f, gi = 0.f;
    if (live) {
      const float re = re_in[fo + f], im = im_in[fo + f];
      const float rr = re + 1e-9f;
      const float mag = sqrtf(rr * rr + im * im);
      float dS = 0.f;
      if (dmel) {
        const int i0 = binmel_idx[2 * f], i1 = binmel_idx[2 * f + 1];
        if (i0 >= 0) dS += binmel_w[2 * f] * dm[i0];
        if (i1 >= 0) dS += binmel_w[2 * f + 1] * dm[i1];
      }
      float dP = 0.f;
      if (dspec) {
        const size_t so = d.spec_T ? (((size_t)b * 2) * d.frames + frame) * F + f : (((size_t)b * 2) * F + f) * d.frames + frame;
        dS += dspec[so] / mag;                            // d log S
        dP = dspec[so + (size_t)F * d.frames] / RTG_PI_REF;
      }
      if (mag > 0.f) {
        gr = dS * rr / mag;
        gi = dS * im / mag;
      }
      const float r2 = re * re + im * im;
      if (dP != 0.f && r2 > 0.f) {
        gr += dP * (-im / r2);
        gi += dP * (re / r2);
      }
      if (f == 0 || f == M) gi = 0.f;                     // structurally-zero imaginary parts carry no gradient
    }
    const float c = (f == 0 || f == M) ? 1.f : 0.5f;      // Hermitian extension: Re z = (z + conj z) / 2
    H[f] = cpx{c * gr, -c * gi};
  }
  __syncthreads();
  // Z'[k] = (H[k] + H[k+M]) + i (H[k] - H[k+M]) W_N^k with H[k+M] = conj H[M-k]
  for (int k = t; k < M; k += TPF) {
    const cpx h0 = H[k], hm = H[M - k];
    const cpx sum{h0.x + hm.x, h0.y - hm.y}, dif{h0.x - hm.x, h0.y + hm.y};
    const cpx w = tw_at(twiddle, N, k);
    const cpx dw = cmul(dif, w);
    A[k] = cpx{sum.x - dw.y, sum.y + dw.x};               // sum + i * dw
  }
  __syncthreads();
  const cpx* Z = fft_half(A, Bf, M, t, TPF, twiddle);          // Z[m] = y[2m] + i y[2m+1]
  if (!live) return;
  float* out = frame_ws + ((size_t)b * d.frames + frame) * d.win;
  for (int n = t; n < d.win; n += TPF) {
    const int i = lpad + n;
    const cpx z = Z[i >> 1];
    out[n] = ((i & 1) ? z.y : z.x) * window[n];
  }
}

__global__ __launch_bounds__(kStftThreads) void stft_bwd_frame_kernel(RtgStftDesc d, const float* __restrict__ re_in,
                                                                      const float* __restrict__ im_in,
                                                                      const float* __restrict__ dmel,
                                                                      const float* __restrict__ dspec,
                                                                      const float* __restrict__ window,
                                                                      const float* __restrict__ twiddle,
                                                                      const int* __restrict__ binmel_idx,
                                                                      const float* __restrict__ binmel_w,
                                                                      float* __restrict__ frame_ws) {
  stft_bwd_frame_body(d, re_in, im_in, dmel, dspec, window, twiddle, binmel_idx, binmel_w, frame_ws, blockIdx.x, blockIdx.y);
}

// sum over the (<= 3) padded positions that alias to sample t of clip b of the frames covering them
__device__ __forceinline__ float stft_ola_at(const RtgStftDesc& d, const float* __restrict__ frame_ws, int b, int t) {
  const int N = d.n_fft, half = N / 2, lpad = (N - d.win) / 2;
  const float* ws = frame_ws + (size_t)b * d.frames * d.win;
  int src[3];
  int ns = 0;
  src[ns++] = t + half;                                   // direct
  if (t >= 1 && t <= half) src[ns++] = half - t;          // left reflection: padded pi < half <- y[half - pi]
  const int pr = 2 * (d.T - 1) - t + half;                // right reflection: padded pi >= T + half
  if (pr >= d.T + half && pr < d.T + N && t < d.T - 1) src[ns++] = pr;
  float acc = 0.f;
  for (int k = 0; k < ns; ++k) {
    const int pi = src[k];
    // frames i with i*hop + lpad <= pi < i*hop + lpad + win
    int hi = (pi - lpad) / d.hop;
    if (pi - lpad < 0) continue;
    int lo = (pi - lpad - d.win + d.hop) / d.hop;        // ceil((pi - lpad - win + 1)/hop)
    if (pi - lpad - d.win + 1 <= 0) lo = 0;
    if (hi > d.frames - 1) hi = d.frames - 1;
    for (int i = lo; i <= hi; ++i) acc += ws[(size_t)i * d.win + (pi - lpad - i * d.hop)];
  }
  return acc;
}

// dy[b,t] += the overlap-add of one resolution
__global__ __launch_bounds__(RTG_THREADS) void stft_ola_kernel(RtgStftDesc d, const float* __restrict__ frame_ws,
                                                               float* __restrict__ dy) {
  const int b = blockIdx.y;
  const int t = blockIdx.x * RTG_THREADS + threadIdx.x;
  if (t >= d.T) return;
  dy[(size_t)b * d.T + t] += stft_ola_at(d, frame_ws, b, t);
}

using StftBwdJob = RtgStftBwdJob;          // (include/rtg.h: descriptor + the operands of rtg_stft_backward but dy)
struct StftBwdJobs {
  int n;
  int blk_end[kStftMaxJobs];
  StftBwdJob j[kStftMaxJobs];
};

__global__ __launch_bounds__(kStftThreads) void stft_bwd_frame_multi_kernel(const StftBwdJobs js) {
  int ji = 0;
  while (ji + 1 < js.n && (int)blockIdx.x >= js.blk_end[ji]) ++ji;
  const StftBwdJob& q = js.j[ji];
  if ((int)blockIdx.y >= q.d.B) return;
  stft_bwd_frame_body(q.d, q.re, q.im, q.dmel, q.dspec, q.window, q.twiddle, q.binmel_idx, q.binmel_w, q.frame_ws,
                      (int)blockIdx.x - (ji ? js.blk_end[ji - 1] : 0), blockIdx.y);
}

// the resolutions of ONE wave [B, T] (every job has the same B, T): dy[b,t] = (accumulate ? dy[b,t] : 0) + the jobs' overlap-adds
// in job order — one thread owns a sample over all resolutions, so the launch needs neither a zeroed dy nor atomics
__global__ __launch_bounds__(RTG_THREADS) void stft_ola_multi_kernel(const StftBwdJobs js, float* __restrict__ dy, int accumulate) {
  const int b = blockIdx.y;
  const int t = blockIdx.x * RTG_THREADS + threadIdx.x;
  const int T = js.j[0].d.T;
  if (t >= T) return;
  float acc = accumulate ? dy[(size_t)b * T + t] : 0.f;
  for (int ji = 0; ji < js.n; ++ji) acc += stft_ola_at(js.j[ji].d, js.j[ji].frame_ws, b, t);
  dy[(size_t)b * T + t] = acc;
}

int validate(const RtgStftDesc* d) {
  if (d->B < 1 || d->T < 2 || d->hop < 1 || d->n_mel < 1 || d->n_mel > 256) return RTG_EINVAL;
  // n_fft = 2 M with M a power of two: min(M / 4, 256) threads own a frame, 8 .. 1 frames per block (torch.stft takes any size;
  // the reference's multi_stft_params, hparam.py:78-80, are 2048 / 1024 / 512)
  if (d->n_fft < 128 || d->n_fft > 4096 || (d->n_fft & (d->n_fft - 1))) return RTG_ERANGE;
  if (d->win < 1 || d->win > d->n_fft) return RTG_EINVAL;
  if (d->frames != 1 + d->T / d->hop) return RTG_EINVAL;
  if (d->n_fft / 2 >= d->T) return RTG_ERANGE;            // reflect padding needs pad < T
  if (d->B > 65535) return RTG_ERANGE;
  return RTG_OK;
}

}  // namespace

extern "C" int rtg_stft_forward(const RtgStftDesc* d, const float* y, const float* window, const float* twiddle,
                                const int* mel_lo, const int* mel_len, const int* mel_woff, const float* mel_w,
                                float* mel, float* spec, float* re, float* im, void* stream) {
  if (!d || !y || !window || !twiddle) return RTG_ENULL;
  int st = validate(d);
  if (st) return st;
  if (mel && (!mel_lo || !mel_len || !mel_woff || !mel_w)) return RTG_ENULL;
  if ((re == nullptr) != (im == nullptr)) return RTG_EINVAL;
  const int M = d->n_fft / 2, fpb = kStftThreads / (M / 4 < kStftThreads ? M / 4 : kStftThreads);
  const size_t lds = (size_t)fpb * (4 * M + M + 1 + 3) * sizeof(float);              // <= 20.5 KB
  RTG_KLAUNCH(stft_fwd_kernel, dim3(rtg_ceil_div(d->frames, fpb), d->B), dim3(kStftThreads), lds, (hipStream_t)stream, *d, y,
              window, twiddle, mel_lo, mel_len, mel_woff, mel_w, mel, spec, re, im);
  return rtg_launch_status();
}

extern "C" int rtg_stft_backward(const RtgStftDesc* d, const float* re, const float* im, const float* dmel,
                                 const float* dspec, const float* window, const float* twiddle, const int* binmel_idx,
                                 const float* binmel_w, float* frame_ws, float* dy, void* stream) {
  if (!d || !re || !im || !window || !twiddle || !frame_ws || !dy) return RTG_ENULL;
  int st = validate(d);
  if (st) return st;
  if (dmel && (!binmel_idx || !binmel_w)) return RTG_ENULL;
  const int M = d->n_fft / 2, fpb = kStftThreads / (M / 4 < kStftThreads ? M / 4 : kStftThreads);
  const size_t lds = (size_t)fpb * (4 * M + 2 * (M + 1) + 256 + 2) * sizeof(float);
  RTG_KLAUNCH(stft_bwd_frame_kernel, dim3(rtg_ceil_div(d->frames, fpb), d->B), dim3(kStftThreads), lds, (hipStream_t)stream, *d,
              re, im, dmel, dspec, window, twiddle, binmel_idx, binmel_w, frame_ws);
  int e = rtg_launch_status();
  if (e) return e;
  RTG_KLAUNCH(stft_ola_kernel, dim3(rtg_ceil_div(d->T, RTG_THREADS), d->B), dim3(RTG_THREADS), 0,
                     (hipStream_t)stream, *d, frame_ws, dy);
  return rtg_launch_status();
}

// ---- several (resolution, signal) jobs in one launch per kernel (ABI 11)
static size_t stft_lds_fwd(const RtgStftDesc* d) {
  const int M = d->n_fft / 2, fpb = kStftThreads / (M / 4 < kStftThreads ? M / 4 : kStftThreads);
  return (size_t)fpb * (4 * M + M + 1 + 3) * sizeof(float);
}
static size_t stft_lds_bwd(const RtgStftDesc* d) {
  const int M = d->n_fft / 2, fpb = kStftThreads / (M / 4 < kStftThreads ? M / 4 : kStftThreads);
  return (size_t)fpb * (4 * M + 2 * (M + 1) + 256 + 2) * sizeof(float);
}
static int stft_fblocks(const RtgStftDesc* d) {
  const int M = d->n_fft / 2, fpb = kStftThreads / (M / 4 < kStftThreads ? M / 4 : kStftThreads);
  return rtg_ceil_div(d->frames, fpb);
}

extern "C" int rtg_stft_forward_multi(int n, const RtgStftFwdJob* jobs, void* stream) {
  if (!jobs) return RTG_ENULL;
  if (n < 1 || n > kStftMaxJobs) return RTG_EINVAL;
  StftFwdJobs js;
  js.n = n;
  size_t lds = 0;
  int end = 0, maxB = 0;
  for (int i = 0; i < n; ++i) {
    const RtgStftFwdJob& q = jobs[i];
    if (!q.y || !q.window || !q.twiddle) return RTG_ENULL;
    const int st = validate(&q.d);
    if (st) return st;
    if (q.mel && (!q.mel_lo || !q.mel_len || !q.mel_woff || !q.mel_w)) return RTG_ENULL;
    if ((q.re == nullptr) != (q.im == nullptr)) return RTG_EINVAL;
    js.j[i] = q;
    end += stft_fblocks(&q.d);
    js.blk_end[i] = end;
    lds = stft_lds_fwd(&q.d) > lds ? stft_lds_fwd(&q.d) : lds;
    maxB = q.d.B > maxB ? q.d.B : maxB;
  }
  for (int i = n; i < kStftMaxJobs; ++i) js.blk_end[i] = end;
  RTG_KLAUNCH(stft_fwd_multi_kernel, dim3(end, maxB), dim3(kStftThreads), lds, (hipStream_t)stream, js);
  return rtg_launch_status();
}

extern "C" int rtg_stft_backward_multi(int n, const RtgStftBwdJob* jobs, float* dy, int accumulate, void* stream) {
  if (!jobs || !dy) return RTG_ENULL;
  if (n < 1 || n > kStftMaxJobs) return RTG_EINVAL;
  StftBwdJobs js;
  js.n = n;
  size_t lds = 0;
  int end = 0;
  for (int i = 0; i < n; ++i) {
    const RtgStftBwdJob& q = jobs[i];
    if (!q.re || !q.im || !q.window || !q.twiddle || !q.frame_ws) return RTG_ENULL;
    const int st = validate(&q.d);
    if (st) return st;
    if (q.dmel && (!q.binmel_idx || !q.binmel_w)) return RTG_ENULL;
    if (q.d.B != jobs[0].d.B || q.d.T != jobs[0].d.T) return RTG_EINVAL;        // the resolutions of ONE wave
    js.j[i] = q;
    end += stft_fblocks(&q.d);
    js.blk_end[i] = end;
    lds = stft_lds_bwd(&q.d) > lds ? stft_lds_bwd(&q.d) : lds;
  }
  for (int i = n; i < kStftMaxJobs; ++i) js.blk_end[i] = end;
  RTG_KLAUNCH(stft_bwd_frame_multi_kernel, dim3(end, jobs[0].d.B), dim3(kStftThreads), lds, (hipStream_t)stream, js);
  int e = rtg_launch_status();
  if (e) return e;
  RTG_KLAUNCH(stft_ola_multi_kernel, dim3(rtg_ceil_div(jobs[0].d.T, RTG_THREADS), jobs[0].d.B), dim3(RTG_THREADS), 0,
              (hipStream_t)stream, js, dy, accumulate);
  return rtg_launch_status();
}
