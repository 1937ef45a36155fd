// Sustained fp32 MFMA rate on this device (dev calibration tool): NACC independent accumulators per wave, WPS waves/SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = threadIdx.x * 0.001f + r;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
void run(int blocks_per_cu, int iters) {
  float* out; hipMalloc(&out, 256 * 1024 * 4 * 8);
  int grid = 256 * blocks_per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, 0.9999f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flop = (double)grid * 4 * iters * 8 * NACC * 4096.0;
    if (rep == 2) printf("NACC=%d blocks/CU=%d iters=%d: %.3f ms  %.1f TF/s\n", NACC, blocks_per_cu, iters, ms, flop / ms / 1e9);
  }
  hipFree(out);
}
int main() {
  run<1>(1, 2000); run<2>(1, 2000); run<4>(1, 2000); run<2>(2, 2000); run<4>(2, 1000); run<2>(4, 1000);
  run<2>(2, 200); run<2>(2, 20000);
  return 0;
}
