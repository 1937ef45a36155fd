// rtg_stft.hip — framed rFFT of get_stft_torch (retunegan/audio.py:150-170) with the |.|, log, angle/PI and mel
// epilogues of multi_stft_loss (retunegan/models/loss.py:32-52) fused, and its backward.
//
// One workgroup (256 threads) per (clip, frame): the win = n_fft/2 windowed samples are gathered (reflect indexing,
// no padded copy of the signal) into an LDS buffer of n_fft complex points, transformed by a radix-2 Stockham
// autosort FFT that ping-pongs between two LDS buffers (twiddles from an fp64-rounded table, L2 resident), and the
// epilogue runs on the n_fft/2+1 bins straight from LDS.  HBM traffic per frame = win floats in, (mel | spec | re,im)
// out: the kernel is output-write bound, the FFT itself never leaves the CU.
// Backward = the same FFT applied to the conjugated cotangent half-spectrum (the adjoint of a real-input DFT),
// windowed into a per-frame workspace, then a gather-form overlap-add that also folds the reflect padding back.
#include "rtg_common.h"

namespace {

#define RTG_PI_REF 3.14159265358979f   // retunegan/utils.py:12

struct cpx {
  float x, y;
};

// In-place-in-LDS radix-2 Stockham FFT of n points.  Returns the buffer that holds the result.
__device__ __forceinline__ cpx* fft_stockham(cpx* a, cpx* b, int n, const float* __restrict__ tw_cos,
                                              const float* __restrict__ tw_sin) {
  const int half = n >> 1;
  cpx *src = a, *dst = b;
  for (int s = 1; s < n; s <<= 1) {          // s = stride, current sub-length = n / s
    // butterfly idx = q + s * p (q < s) takes src[idx] and src[idx + n / 2] to dst[q + 2 s p] and dst[q + 2 s p + s] with
    // the twiddle exp(-2 pi i p / (n / s)) = table[s p]: s p = idx with its low log2(s) bits cleared — masks, no division
    // (measured: the same 23 us per launch as with the divisions, and as with the twiddle table staged in LDS — the passes'
    // barriers and LDS round trips set the time, not their arithmetic)
    const int hi = ~(s - 1);
#pragma unroll 4
    for (int idx = threadIdx.x; idx < half; idx += RTG_THREADS) {
      const int k = idx & hi;
      const cpx u = src[idx], v = src[idx + half];
      const float c = tw_cos[k], sn = tw_sin[k];
      const float dx = u.x - v.x, dy = u.y - v.y;
      cpx o0, o1;
      o0.x = u.x + v.x; o0.y = u.y + v.y;
      o1.x = dx * c + dy * sn;                // (dx + i dy) * (c - i sn)
      o1.y = dy * c - dx * sn;
      dst[idx + k] = o0;
      dst[idx + k + s] = o1;
    }
    __syncthreads();
    cpx* t = src; src = dst; dst = t;
  }
  return src;
}

__global__ __launch_bounds__(RTG_THREADS) void stft_fwd_kernel(RtgStftDesc d, const float* __restrict__ y,
                                                               const float* __restrict__ window,
                                                               const float* __restrict__ twiddle,
                                                               const int* __restrict__ mel_lo,
                                                               const int* __restrict__ mel_len,
                                                               const int* __restrict__ mel_woff,
                                                               const float* __restrict__ mel_w, float* mel, float* spec,
                                                               float* re_out, float* im_out) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int N = d.n_fft, F = N / 2 + 1;
  cpx* A = reinterpret_cast<cpx*>(smem);
  cpx* Bf = A + N;
  float* S = reinterpret_cast<float*>(Bf + N);          // F magnitudes
  const int frame = blockIdx.x, b = blockIdx.y;
  const int lpad = (N - d.win) / 2;
  const float* yb = y + (size_t)b * d.T;

  for (int i = threadIdx.x; i < N; i += RTG_THREADS) {
    float v = 0.f;
    const int n = i - lpad;
    if (n >= 0 && n < d.win) {
      int t = frame * d.hop + i - N / 2;                 // centre=True: padded index - n_fft/2
      if (t < 0) t = -t;
      if (t >= d.T) t = 2 * (d.T - 1) - t;
      v = yb[t] * window[n];
    }
    A[i].x = v;
    A[i].y = 0.f;
  }
  __syncthreads();
  const cpx* X = fft_stockham(A, Bf, N, twiddle, twiddle + N / 2);

  const size_t fo = ((size_t)b * d.frames + frame) * F;  // [B][frames][F] scratch layout for the backward
  for (int f = threadIdx.x; f < F; f += RTG_THREADS) {
    // DC and Nyquist of a real-input transform are exactly real: a real-to-complex FFT (torch.stft / pocketfft) returns
    // imag = +0.0 there, so angle() is exactly 0 or +pi; rounding noise of a complex FFT would flip it to -pi at random
    const float re = X[f].x, im = (f == 0 || f == N / 2) ? 0.f : X[f].y;
    const float rr = re + 1e-9f;
    const float mag = sqrtf(rr * rr + im * im);
    S[f] = mag;
    if (re_out) {
      re_out[fo + f] = re;
      im_out[fo + f] = im;
    }
    if (spec) {
      // [B][2][F][frames] (the reference's stack, loss.py:36-44: lanes 4 bytes x `frames` apart) or, spec_T, [B][2][frames][F]:
      // a frame's bins are consecutive — coalesced stores, and the layout the spectrogram discriminators walk (round 5)
      const size_t so = d.spec_T ? (((size_t)b * 2) * d.frames + frame) * F + f : (((size_t)b * 2) * F + f) * d.frames + frame;
      spec[so] = logf(mag);
      spec[so + (size_t)F * d.frames] = atan2f(im, re) / RTG_PI_REF;
    }
  }
  __syncthreads();
  if (mel) {
    for (int m = threadIdx.x; m < d.n_mel; m += RTG_THREADS) {
      const int lo = mel_lo[m], len = mel_len[m];
      const float* w = mel_w + mel_woff[m];
      float acc = 0.f;
      for (int i = 0; i < len; ++i) acc += w[i] * S[lo + i];
      mel[((size_t)b * d.n_mel + m) * d.frames + frame] = acc;
    }
  }
}

__global__ __launch_bounds__(RTG_THREADS) void stft_bwd_frame_kernel(RtgStftDesc d, const float* __restrict__ re_in,
                                                                     const float* __restrict__ im_in,
                                                                     const float* __restrict__ dmel,
                                                                     const float* __restrict__ dspec,
                                                                     const float* __restrict__ window,
                                                                     const float* __restrict__ twiddle,
                                                                     const int* __restrict__ binmel_idx,
                                                                     const float* __restrict__ binmel_w,
                                                                     float* __restrict__ frame_ws) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int N = d.n_fft, F = N / 2 + 1;
  cpx* A = reinterpret_cast<cpx*>(smem);
  cpx* Bf = A + N;
  float* dm = reinterpret_cast<float*>(Bf + N);          // n_mel cotangents of this frame
  const int frame = blockIdx.x, b = blockIdx.y;
  const int lpad = (N - d.win) / 2;

  if (dmel)
    for (int m = threadIdx.x; m < d.n_mel; m += RTG_THREADS)
      dm[m] = dmel[((size_t)b * d.n_mel + m) * d.frames + frame];
  __syncthreads();
  const size_t fo = ((size_t)b * d.frames + frame) * F;
  for (int f = threadIdx.x; f < N; f += RTG_THREADS) {
    float gr = 0.f, gi = 0.f;
    if (f < F) {
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
      if (f == 0 || f == N / 2) gi = 0.f;                 // structurally-zero imaginary parts carry no gradient
    }
    A[f].x = gr;                                          // conj(G): adjoint of the forward DFT = Re FFT(conj G)
    A[f].y = -gi;
  }
  __syncthreads();
  const cpx* Z = fft_stockham(A, Bf, N, twiddle, twiddle + N / 2);
  float* out = frame_ws + ((size_t)b * d.frames + frame) * d.win;
  for (int n = threadIdx.x; n < d.win; n += RTG_THREADS) out[n] = Z[lpad + n].x * window[n];
}

// dy[b,t] += sum over the (<= 3) padded positions that alias to t of the frames covering them.
__global__ __launch_bounds__(RTG_THREADS) void stft_ola_kernel(RtgStftDesc d, const float* __restrict__ frame_ws,
                                                               float* __restrict__ dy) {
  const int b = blockIdx.y;
  const int t = blockIdx.x * RTG_THREADS + threadIdx.x;
  if (t >= d.T) return;
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
  dy[(size_t)b * d.T + t] += acc;
}

int validate(const RtgStftDesc* d) {
  if (d->B < 1 || d->T < 2 || d->hop < 1 || d->n_mel < 1 || d->n_mel > 256) return RTG_EINVAL;
  if (d->n_fft != 512 && d->n_fft != 1024 && d->n_fft != 2048 && d->n_fft != 256 && d->n_fft != 4096) return RTG_ERANGE;
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
  const size_t lds = (size_t)d->n_fft * 2 * sizeof(cpx) + (size_t)(d->n_fft / 2 + 1) * sizeof(float);
  if (lds > 64 * 1024)
    hipFuncSetAttribute((const void*)stft_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  RTG_KLAUNCH(stft_fwd_kernel, dim3(d->frames, d->B), dim3(RTG_THREADS), lds, (hipStream_t)stream, *d, y, window,
                     twiddle, mel_lo, mel_len, mel_woff, mel_w, mel, spec, re, im);
  return rtg_launch_status();
}

extern "C" int rtg_stft_backward(const RtgStftDesc* d, const float* re, const float* im, const float* dmel,
                                 const float* dspec, const float* window, const float* twiddle, const int* binmel_idx,
                                 const float* binmel_w, float* frame_ws, float* dy, void* stream) {
  if (!d || !re || !im || !window || !twiddle || !frame_ws || !dy) return RTG_ENULL;
  int st = validate(d);
  if (st) return st;
  if (dmel && (!binmel_idx || !binmel_w)) return RTG_ENULL;
  const size_t lds = (size_t)d->n_fft * 2 * sizeof(cpx) + 256 * sizeof(float);
  if (lds > 64 * 1024)
    hipFuncSetAttribute((const void*)stft_bwd_frame_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  RTG_KLAUNCH(stft_bwd_frame_kernel, dim3(d->frames, d->B), dim3(RTG_THREADS), lds, (hipStream_t)stream, *d, re,
                     im, dmel, dspec, window, twiddle, binmel_idx, binmel_w, frame_ws);
  int e = rtg_launch_status();
  if (e) return e;
  RTG_KLAUNCH(stft_ola_kernel, dim3(rtg_ceil_div(d->T, RTG_THREADS), d->B), dim3(RTG_THREADS), 0,
                     (hipStream_t)stream, *d, frame_ws, dy);
  return rtg_launch_status();
}
