// MFMA experiment: can the constant-modulus half of a Montgomery multiplication -- m = T_lo p' mod R and
// m p, 182 of the 351 v_mad_u64_u32 of one 13 x 30-bit multiplication, 1521 of the 3055 per bucket addition -- move from the
// VALU to the matrix cores?  For the 64 buckets of a wave, m p is a (64 x 48 byte) x Toeplitz(p) (48 x 96 byte) product:
// v_mfma_i32_16x16x64_i8 does 16 x 16 x 64 byte products per instruction, so 4 row tiles x 3 column tiles cover one
// constant product (12 MFMAs; 24 for the two of a reduction).  KILL CRITERION (stated before measuring): keep only if a
// bucket addition could drop by >= 15 %, i.e. if the MFMA route reduces faster than ~1.3 x the VALU's own rate for the
// reduction part (f30_mul: 78 G/s whole multiplications = ~150 G/s "reduction-only" equivalents).
//
// What this program measures is an UPPER BOUND of the MFMA route's rate: the data movement and the MFMAs of one reduction,
// WITHOUT the carry logic (the signed-byte conversion of the operands and the recombination of 96 byte-columns of i32 into
// limbs, ~250 more VALU instructions per lane) -- if even this skeleton is slower than the VALU, the route is dead:
//   per lane and constant product:  operand bytes lane-major -> LDS (the MFMA's A operand wants bucket i's bytes spread over
//   lanes i, i + 16, i + 32, i + 48; the accumulate kernel has one bucket per lane), 4 x ds_read_b128 of A tiles, 12 MFMAs
//   against register-resident Toeplitz tiles, 48 i32 column sums per bucket back through LDS into bucket-per-lane order.
// Variants: full (LDS both ways), nolds_out (column sums stay in MFMA layout: what a carry logic IN that layout would see),
// mfma_only (no LDS at all: the matrix-core rate itself).
//
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_reduce_bench.hip -o tools/mfma_reduce_bench && tools/mfma_reduce_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef int v4i __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// MODE 0: full, 1: no LDS on the way out, 2: MFMAs only
template <int MODE>
__global__ __launch_bounds__(256) void reduce_skeleton(int* __restrict__ out, int iters) {
  __shared__ __attribute__((aligned(16))) int lds[4][64 * 52];      // per wave: 64 buckets x (48 + 4 pad) dwords
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int* L = lds[wave];
  // operand of this lane's bucket: 12 dwords (48 bytes)
  v4i op[3];
  for (int k = 0; k < 3; k++) op[k] = v4i{lane * 7 + k, lane * 13 + k, lane * 17 + k, lane * 19 + k};
  // Toeplitz(p) column tiles as B operands (constants in registers): 3 tiles x 16 bytes per lane
  v4i bt[3];
  for (int k = 0; k < 3; k++) bt[k] = v4i{0x01020304 * (k + 1), 0x11121314, 0x21222324 + lane, 0x31323334};
  int sink = 0;
  for (int it = 0; it < iters; it++) {
    for (int prod = 0; prod < 2; prod++) {                          // m = T_lo p' ; then m p
      if (MODE != 2) {
        // bucket-per-lane -> LDS, lane-major rows of 13 x 16 B (padded): 3 x ds_write_b128
#pragma unroll
        for (int k = 0; k < 3; k++) *reinterpret_cast<v4i*>(L + lane * 52 + 4 * k) = op[k];
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
      }
      v4i acc[4][3];
#pragma unroll
      for (int rt = 0; rt < 4; rt++) {                              // row tile = 16 buckets
        v4i a;
        if (MODE != 2) a = *reinterpret_cast<const v4i*>(L + (rt * 16 + (lane & 15)) * 52 + 4 * (lane >> 4));   // bucket rt*16 + lane%16, k-block lane/16 (3 of 4 blocks carry data)
        else a = op[rt % 3];
#pragma unroll
        for (int ct = 0; ct < 3; ct++) acc[rt][ct] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, bt[ct], v4i{0, 0, 0, 0}, 0, 0, 0);
      }
      if (MODE == 0) {
        // 48 column sums per bucket back to bucket-per-lane order: D[rt][ct] holds rows 4 (lane / 16) .. + 3 (buckets) of column lane % 16
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int rt = 0; rt < 4; rt++)
#pragma unroll
          for (int ct = 0; ct < 3; ct++)
#pragma unroll
            for (int r = 0; r < 4; r++) L[(rt * 16 + 4 * (lane >> 4) + r) * 52 + ct * 16 + (lane & 15)] = acc[rt][ct][r];
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
        v4i col[12];
#pragma unroll
        for (int k = 0; k < 12; k++) col[k] = *reinterpret_cast<const v4i*>(L + lane * 52 + 4 * k);
        // stand-in for the carry logic: fold the 48 columns into the next operand (the real one is ~250 VALU instructions)
#pragma unroll
        for (int k = 0; k < 3; k++) op[k] = col[k] + col[k + 3] + col[k + 6] + col[k + 9];
      } else {
#pragma unroll
        for (int k = 0; k < 3; k++) op[k] = acc[0][k] + acc[1][k] + acc[2][k] + acc[3][k];
      }
    }
    sink += op[0][0];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = sink + op[1][1] + op[2][2];
}

template <int MODE>
int run(const char* name, int* d_out, int blocks, int iters) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL(reduce_skeleton<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, 8);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  hipLaunchKernelGGL(reduce_skeleton<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, iters);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  const double red = (double)blocks * 256 * iters;
  const double mfma = (double)blocks * 4 * iters * 24;
  printf("%-10s %8.3f ms   %7.1f G reductions/s (lane reductions; VALU baseline ~150 G/s reduction-only, 78 G/s whole multiplications)   %6.2f T MFMA/s = %.0f TOPS i8\n",
         name, ms, red / ms / 1e6, mfma / ms / 1e9, mfma * 16 * 16 * 64 * 2 / ms / 1e9);
  return 0;
}

int main() {
  int dev; CK(hipGetDevice(&dev));
  hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, dev));
  const int blocks = pr.multiProcessorCount * 8, iters = 2000;
  int* d_out; CK(hipMalloc(&d_out, (size_t)blocks * 256 * 4));
  printf("%s, %d CUs; one 'reduction' = the two constant products of a Montgomery reduction of a 384-bit value (24 x v_mfma_i32_16x16x64_i8 per wave of 64 buckets), carry logic NOT included\n", pr.gcnArchName, pr.multiProcessorCount);
  if (run<2>("mfma_only", d_out, blocks, iters)) return 1;
  if (run<1>("nolds_out", d_out, blocks, iters)) return 1;
  if (run<0>("full", d_out, blocks, iters)) return 1;
  return 0;
}
