// Dev calibration: which ingredient of the conv inner loop costs MFMA throughput?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MF(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0)

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, const float* w, int iters, int row) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 16 * row; i += 256) lds[i] = i * 1e-4f;
  __syncthreads();
  f32x16 acc0, acc1;
  for (int r = 0; r < 16; ++r) { acc0[r] = r; acc1[r] = -r; }
  float av[8], bv[8][2];
  for (int c = 0; c < 8; ++c) { av[c] = 1.f + c * 1e-3f + lane * 1e-5f; bv[c][0] = 0.5f + c * 1e-3f; bv[c][1] = 0.25f - c * 1e-3f; }
  const float* bp = lds + (lane >> 5) * row + (lane & 31);
  const float* wp = w + lane;
  float an[8];
  for (int it = 0; it < iters; ++it) {
    if (MODE >= 3) {
#pragma unroll
      for (int c = 0; c < 8; ++c) an[c] = wp[(size_t)((it + 1) & 63) * 512 + c * 64];
    }
    if (MODE >= 2) {
      const float* b2 = bp + (it & 7);
#pragma unroll
      for (int c = 0; c < 8; ++c) { bv[c][0] = b2[c * 2 * row]; bv[c][1] = b2[c * 2 * row + 32]; }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) { acc0 = MF(av[c], bv[c][0], acc0); acc1 = MF(av[c], bv[c][1], acc1); }
    if (MODE >= 2) __builtin_amdgcn_sched_barrier(0);
    if (MODE >= 3) {
#pragma unroll
      for (int c = 0; c < 8; ++c) av[c] = an[c];
    }
  }
  float s = 0;
  for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE>
void run(int bpc, int iters) {
  float *out, *w; hipMalloc(&out, 256 * 8 * 1024 * 4); hipMalloc(&w, 64 * 512 * 4 + 4096); hipMemset(w, 0, 64 * 512 * 4 + 4096);
  int grid = 256 * bpc, row = 304;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 16 * row * 4, 0, out, w, iters, row);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  double flop = (double)grid * 4 * iters * 16 * 4096.0;
  printf("MODE=%d blocks/CU=%d iters=%d: %.3f ms  %.1f TF/s\n", MODE, bpc, iters, ms, flop / ms / 1e9);
  hipFree(out); hipFree(w);
}
int main() {
  for (int bpc = 1; bpc <= 3; ++bpc) { run<1>(bpc, 400); run<2>(bpc, 400); run<3>(bpc, 400); }
  return 0;
}
