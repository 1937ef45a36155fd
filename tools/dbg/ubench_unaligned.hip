// ubench_unaligned.hip — what does a 16-byte buffer load cost at 2-byte alignment?  (bf16 NCW rows of odd length start at
// 2-byte aligned addresses; the bf16 staging of rtg_dconv.hip wants one 16-byte load per 8 positions of a channel.)
//   hipcc --offload-arch=gfx950 -O3 tools/dbg/ubench_unaligned.hip -o /tmp/ub && /tmp/ub
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

using rsrc_t = __amdgpu_buffer_rsrc_t;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// every lane loads 16 bytes at  row * pitch_bytes + lane16 * 16 + mis  (rows of `pitch_bytes`), sums them
template <int UNROLL>
__global__ void k_load(const unsigned short* p, unsigned bytes, unsigned pitch_bytes, unsigned mis, unsigned rows,
                       unsigned per_row, unsigned* out) {
  const rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, bytes, 0x00020000);
  unsigned acc = 0;
  const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x, nth = gridDim.x * blockDim.x;
  const unsigned total = rows * per_row;
  for (unsigned i = tid; i < total; i += nth * UNROLL) {
    u32x4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const unsigned j = i + u * nth;
      const unsigned row = j / per_row, c = j - row * per_row;
      const unsigned off = j < total ? row * pitch_bytes + c * 16u + mis : 0x80000000u;
      v[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) acc += v[u][0] ^ v[u][1] ^ v[u][2] ^ v[u][3];
  }
  if (acc == 0x12345678u) out[0] = acc;
}

// correctness: lane loads 16 bytes at byte offset `off0 + 2 * lane` and writes them out
__global__ void k_check(const unsigned short* p, unsigned bytes, unsigned off0, unsigned short* out) {
  const rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, bytes, 0x00020000);
  const unsigned off = off0 + 2u * threadIdx.x;
  u32x4 v = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
  for (int i = 0; i < 4; ++i) {
    out[threadIdx.x * 8 + 2 * i] = (unsigned short)(v[i] & 0xffff);
    out[threadIdx.x * 8 + 2 * i + 1] = (unsigned short)(v[i] >> 16);
  }
}

int main() {
  const unsigned rows = 1 << 16, pitch_elems = 2048 + 14;                 // rows of 2062 bf16: 4124 bytes (4-byte aligned rows)
  const size_t n = (size_t)rows * pitch_elems + 64;
  std::vector<unsigned short> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = (unsigned short)(i * 2654435761u >> 7);
  unsigned short* d; unsigned* o; unsigned short* oc;
  hipMalloc(&d, n * 2); hipMalloc(&o, 64); hipMalloc(&oc, 64 * 8 * 2);
  hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice);
  // ---- correctness at 2-byte alignment, including the tail of the buffer (out of range -> zeros per dword?)
  for (unsigned off0 : {0u, 2u, 6u, 14u}) {
    k_check<<<1, 64>>>(d, (unsigned)(n * 2), off0, oc);
    std::vector<unsigned short> r(512);
    hipMemcpy(r.data(), oc, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
      for (int e = 0; e < 8; ++e)
        if (r[l * 8 + e] != h[off0 / 2 + l + e]) ++bad;
    printf("check off0=%u: %d mismatches of 512\n", off0, bad);
  }
  {  // range check at the end of the buffer: num_records = 100 bytes, lane loads at 2 * lane
    k_check<<<1, 64>>>(d, 100u, 0u, oc);
    std::vector<unsigned short> r(512);
    hipMemcpy(r.data(), oc, 1024, hipMemcpyDeviceToHost);
    printf("range check (num_records = 100 bytes = 50 elements): lane: elements read (non-zero = in range value)\n");
    for (int l = 38; l < 52; ++l) {
      printf("  lane %d (elements %d..%d):", l, l, l + 7);
      for (int e = 0; e < 8; ++e) printf(" %s", r[l * 8 + e] == h[l + e] ? "ok" : (r[l * 8 + e] == 0 ? "0" : "??"));
      printf("\n");
    }
  }
  // ---- throughput: rows of 4124 bytes, 256 16-byte loads per row, misalignment 0 / 4 / 2 / 6 / 10 (rows themselves shift the
  // alignment by 12 bytes per row: 4124 % 16 = 12)
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (unsigned pitch : {4096u, 4124u, 4126u})
    for (unsigned mis : {0u, 4u, 2u, 6u, 10u}) {
      const unsigned per_row = 256, bytes = (unsigned)(n * 2);
      k_load<4><<<256 * 8, 256>>>(d, bytes, pitch, mis, rows, per_row, o);
      hipEventRecord(e0);
      for (int it = 0; it < 5; ++it) k_load<4><<<256 * 8, 256>>>(d, bytes, pitch, mis, rows, per_row, o);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("pitch %u mis %2u: %.1f GB/s\n", pitch, mis, 5.0 * rows * per_row * 16 / (ms * 1e-3) / 1e9);
    }
  return 0;
}
