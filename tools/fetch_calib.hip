// FETCH_SIZE calibration on the access pattern of msmfb::accum30_kernel.
//
// MI355X_MICROARCH.md calibrates the gfx950 FETCH_SIZE under-count (x2) for 16 B/lane COALESCED streams and calls other
// widths "uncalibrated".  The accumulate kernel does neither: every lane gathers ONE 128-byte table point (two 64-byte
// coordinates, 8 x global_load_dwordx4) from a pseudo-random slot of a multi-gigabyte table.  This program issues exactly
// that pattern with a KNOWN byte count, plus a coalesced 16 B/lane stream of the same size as a control, so that
//     factor = known bytes / (FETCH_SIZE KiB x 1024)
// can be read off a `rocprofv3 --pmc FETCH_SIZE` capture of it (tools/fetch_calib.sh) and applied by tools/pmc_summary.py
// to the accumulate kernel instead of the stream factor.
//
//   hipcc --offload-arch=gfx950 -O3 tools/fetch_calib.hip -o tools/fetch_calib
//   tools/fetch_calib [table MiB = 2048] [gathers = 2^24]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

struct Pt { uint32_t x[16]; uint32_t y[16]; };   // msmfb::G1Aff30: 2 x 64 B

__device__ __forceinline__ uint32_t mix(uint32_t h) { h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16; return h; }

// one 128-byte point per lane from a pseudo-random slot: bytes fetched = gathers x 128 (every slot distinct lines)
__global__ __launch_bounds__(256) void gather128_kernel(const Pt* __restrict__ tab, uint64_t slots, uint64_t n, uint32_t* __restrict__ sink) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t s = (((uint64_t)mix((uint32_t)i) << 32) | mix((uint32_t)i ^ 0xdeadbeefu)) % slots;     // every lane its own slot
  const uint4* p = reinterpret_cast<const uint4*>(tab + s);
  uint32_t acc = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) { const uint4 q = p[k]; acc ^= q.x + q.y + q.z + q.w; }
  if (acc == 0x12345678u) sink[0] = acc;          // never true for the fill pattern; keeps the loads alive
}
// the same gather, list-driven like the real kernel: lane reads a 4-byte index (coalesced) and then the point
__global__ __launch_bounds__(256) void gather128_list_kernel(const Pt* __restrict__ tab, const uint32_t* __restrict__ idx, uint64_t n,
                                                            uint32_t* __restrict__ sink) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint4* p = reinterpret_cast<const uint4*>(tab + idx[i]);
  uint32_t acc = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) { const uint4 q = p[k]; acc ^= q.x + q.y + q.z + q.w; }
  if (acc == 0x12345678u) sink[0] = acc;
}
// control: coalesced stream, 16 B per lane (the pattern the guide's x2 was calibrated on)
__global__ __launch_bounds__(256) void stream16_kernel(const uint4* __restrict__ src, uint64_t n16, uint32_t* __restrict__ sink) {
  uint32_t acc = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint4 q = src[i]; acc ^= q.x + q.y + q.z + q.w;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}
__global__ void fill_kernel(uint32_t* p, uint64_t words) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (uint64_t)gridDim.x * blockDim.x) p[i] = (uint32_t)i * 2654435761u | 1u;
}
__global__ void fill_idx_kernel(uint32_t* idx, uint64_t n, uint64_t slots) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) idx[i] = (uint32_t)((((uint64_t)mix((uint32_t)i * 3u + 1u) << 20) ^ mix((uint32_t)i + 77u)) % slots);
}

int main(int argc, char** argv) {
  const uint64_t mib = argc > 1 ? strtoull(argv[1], 0, 10) : 2048;
  const uint64_t n = argc > 2 ? strtoull(argv[2], 0, 10) : (1ull << 24);
  const uint64_t bytes = mib << 20, slots = bytes / sizeof(Pt);
  Pt* tab; uint32_t* sink; uint32_t* idx;
  CK(hipMalloc(&tab, bytes)); CK(hipMalloc(&sink, 64)); CK(hipMalloc(&idx, n * 4));
  hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, (uint32_t*)tab, bytes / 4);
  hipLaunchKernelGGL(fill_idx_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, idx, n, slots);
  CK(hipDeviceSynchronize());
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float ms;
  for (int rep = 0; rep < 2; rep++) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(gather128_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, tab, slots, n, sink);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
    printf("gather128_kernel       known_bytes %llu  %.3f ms  %.1f GB/s\n", (unsigned long long)(n * 128), ms, n * 128 / ms / 1e6);
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(gather128_list_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, tab, idx, n, sink);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
    printf("gather128_list_kernel  known_bytes %llu  %.3f ms  %.1f GB/s\n", (unsigned long long)(n * 132), ms, n * 132 / ms / 1e6);
    const uint64_t n16 = (n * 128) / 16;
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(stream16_kernel, dim3(8192), dim3(256), 0, 0, (const uint4*)tab, n16, sink);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
    printf("stream16_kernel        known_bytes %llu  %.3f ms  %.1f GB/s\n", (unsigned long long)(n16 * 16), ms, n16 * 16 / ms / 1e6);
  }
  return 0;
}
