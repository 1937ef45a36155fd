// rtg_common.h — shared device/host helpers for librtg.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/rtg.h"

#define RTG_CK 16            // input channels staged per LDS chunk
#define RTG_PW_MAX 576       // widest input patch (floats per channel row) a block stages
#define RTG_THREADS 256      // 4 wavefronts of 64

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Every kernel launch of the library: hipGetLastError() is per host thread and sticky, so an error left behind by an
// unrelated earlier runtime call (e.g. a device query made before the context existed) would be reported as the
// status of our launch.  Clear it first; what rtg_launch_status() sees afterwards belongs to this launch.
#define RTG_KLAUNCH(...)              \
  do {                                \
    (void)hipGetLastError();          \
    hipLaunchKernelGGL(__VA_ARGS__);  \
  } while (0)

static inline int rtg_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? RTG_OK : -(1000 + (int)e);
}

// A/B and tuning knobs of the host launch path are read from the environment ONCE per process (a conv launch must not
// scan the environment; changing a knob mid-process would also silently invalidate the choices cached in rtg/tune.py).
#include <stdlib.h>
#define RTG_ENV_SET(name) ([]() -> bool { static const bool v = getenv(name) != nullptr; return v; }())
#define RTG_ENV_INT(name, dflt) \
  ([]() -> int { static const int v = []() -> int { const char* e = getenv(name); return e ? atoi(e) : (dflt); }(); return v; }())

// The > 64 KB dynamic-LDS opt-in of a kernel.  hipFuncSetAttribute applies to the CURRENT device, so it is tracked per
// device (a bit per ordinal) in an atomic the call site owns — one `static std::atomic<unsigned>` per kernel instance; safe
// from several host threads (the forked autograd threads launch concurrently: setting the attribute twice is harmless).
#include <atomic>
static inline int rtg_lds_optin(const void* kernel, std::atomic<unsigned>& done) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return RTG_ERANGE;
  const unsigned bit = 1u << (dev & 31);
  if (!(done.load(std::memory_order_acquire) & bit)) {
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return RTG_ERANGE;
    done.fetch_or(bit, std::memory_order_release);
  }
  return RTG_OK;
}

static inline int rtg_ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }

__device__ __forceinline__ float rtg_lrelu(float v, float slope) { return v > 0.f ? v : v * slope; }

// sum across the 64 lanes of a wavefront (result valid in every lane)
__device__ __forceinline__ float rtg_wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
// block-wide sum for RTG_THREADS threads; `red` is a 4-float LDS scratch. Result valid in every thread.
__device__ __forceinline__ float rtg_block_sum(float v, float* red) {
  v = rtg_wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}
