// Instruction-rate and field-multiplication microbenchmarks for gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I marlin_amd/csrc tools/microbench.hip -o tools/microbench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "ff.cuh"
#include "g1.cuh"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 2048;

__global__ void k_mad64(u64* out, u32 a, u32 b) {
  u64 acc[8]; for (int i = 0; i < 8; i++) acc[i] = threadIdx.x + i;
  u32 x = a + threadIdx.x, y = b;
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(x), "v"(y) : "vcc");
  }
  u64 s = 0; for (int i = 0; i < 8; i++) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_mullo(u64* out, u32 a, u32 b) {
  u32 acc[8]; for (int i = 0; i < 8; i++) acc[i] = threadIdx.x + i + a;
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(acc[i]) : "v"(b));
  }
  u64 s = 0; for (int i = 0; i < 8; i++) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_mulhi(u64* out, u32 a, u32 b) {
  u32 acc[8]; for (int i = 0; i < 8; i++) acc[i] = threadIdx.x + i + a;
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(acc[i]) : "v"(b));
  }
  u64 s = 0; for (int i = 0; i < 8; i++) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_addc(u64* out, u32 a, u32 b) {
  u32 acc[8]; for (int i = 0; i < 8; i++) acc[i] = threadIdx.x + i + a;
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) asm volatile("v_addc_co_u32 %0, vcc, %1, %0, vcc" : "+v"(acc[i]) : "v"(b) : "vcc");
  }
  u64 s = 0; for (int i = 0; i < 8; i++) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_mad24(u64* out, u32 a, u32 b) {
  u32 acc[8]; for (int i = 0; i < 8; i++) acc[i] = threadIdx.x + i + a;
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(acc[i]) : "v"(b));
  }
  u64 s = 0; for (int i = 0; i < 8; i++) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_madpair(u64* out, u32 a, u32 b) {   // mad + addc pairs as in the Comba column
  u64 acc[4]; u32 hi[4]; for (int i = 0; i < 4; i++) { acc[i] = threadIdx.x + i; hi[i] = 0; }
  u32 x = a + threadIdx.x, y = b;
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < 4; i++)
      asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %3, %2, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"
                   : "+v"(acc[i]), "+v"(hi[i]) : "v"(x), "v"(y) : "vcc");
  }
  u64 s = 0; for (int i = 0; i < 4; i++) s += acc[i] + hi[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// ---- controls: which VALU classes issue a wave64 instruction in 2 cycles (32 lanes / clock,
// MI355X_MICROARCH.md "v_fma_f32 2 cyc") and which in 4 (16 lanes / clock)?  Same harness as above: 8 independent chains per
// thread, 8 waves per SIMD, ITERS x 8 instructions per thread; every kernel also reports shader cycles per
// wave-instruction from s_memtime, which is independent of the clock the chip happens to run at.
__device__ __forceinline__ u64 memtime() { u64 t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)); return t; }
#define CTRL_KERNEL32(NAME, ASM, INIT)                                                                         \
  __global__ void NAME(u64* out, u32 a, u32 b, unsigned long long* cyc) {                                    \
    u32 acc[8]; for (int i = 0; i < 8; i++) acc[i] = INIT;                                                     \
    const u64 t0 = memtime();                                                                                  \
    for (int it = 0; it < ITERS; it++) {                                                                       \
      _Pragma("unroll") for (int i = 0; i < 8; i++) asm volatile(ASM : "+v"(acc[i]) : "v"(b), "v"(a));         \
    }                                                                                                          \
    const u64 t1 = memtime();                                                                                  \
    if ((threadIdx.x & 63) == 0) atomicAdd(cyc, (unsigned long long)(t1 - t0));                                \
    u64 s = 0; for (int i = 0; i < 8; i++) s += acc[i];                                                        \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                            \
  }
CTRL_KERNEL32(k_c_fma_f32, "v_fma_f32 %0, %0, %1, %2", __int_as_float(0x3f800000 + threadIdx.x + i))
CTRL_KERNEL32(k_c_add_u32, "v_add_u32 %0, %0, %1", threadIdx.x + i + a)
CTRL_KERNEL32(k_c_and_b32, "v_and_b32 %0, %0, %1", threadIdx.x + i + a)
CTRL_KERNEL32(k_c_lshrrev_b32, "v_lshrrev_b32 %0, 1, %0", threadIdx.x + i + a)
CTRL_KERNEL32(k_c_and_or_b32, "v_and_or_b32 %0, %0, %1, %2", threadIdx.x + i + a)
CTRL_KERNEL32(k_c_add3_u32, "v_add3_u32 %0, %0, %1, %2", threadIdx.x + i + a)
CTRL_KERNEL32(k_c_mad_u32_u24, "v_mad_u32_u24 %0, %0, %1, %2", threadIdx.x + i + a)
CTRL_KERNEL32(k_c_mul_lo_u32, "v_mul_lo_u32 %0, %0, %1", threadIdx.x + i + a)
CTRL_KERNEL32(k_c_addc_co_u32, "v_addc_co_u32 %0, vcc, %1, %0, vcc", threadIdx.x + i + a)
CTRL_KERNEL32(k_c_add_co_u32, "v_add_co_u32 %0, vcc, %1, %0", threadIdx.x + i + a)
#define CTRL_KERNEL64(NAME, ASM)                                                                               \
  __global__ void NAME(u64* out, u32 a, u32 b, unsigned long long* cyc) {                                    \
    u64 acc[8]; for (int i = 0; i < 8; i++) acc[i] = ((u64)(threadIdx.x + i) << 33) + a;                       \
    const u32 x = a + threadIdx.x;                                                                             \
    const u64 t0 = memtime();                                                                                  \
    for (int it = 0; it < ITERS; it++) {                                                                       \
      _Pragma("unroll") for (int i = 0; i < 8; i++) asm volatile(ASM : "+v"(acc[i]) : "v"(x), "v"(b) : "vcc"); \
    }                                                                                                          \
    const u64 t1 = memtime();                                                                                  \
    if ((threadIdx.x & 63) == 0) atomicAdd(cyc, (unsigned long long)(t1 - t0));                                \
    u64 s = 0; for (int i = 0; i < 8; i++) s += acc[i];                                                        \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                            \
  }
CTRL_KERNEL64(k_c_mad_u64_u32, "v_mad_u64_u32 %0, vcc, %1, %2, %0")
CTRL_KERNEL64(k_c_lshrrev_b64, "v_lshrrev_b64 %0, 30, %0")
CTRL_KERNEL64(k_c_lshl_add_u64, "v_lshl_add_u64 %0, %0, 0, %0")
CTRL_KERNEL64(k_c_pk_fma_f32, "v_pk_fma_f32 %0, %0, %0, %0")
CTRL_KERNEL64(k_c_fma_f64, "v_fma_f64 %0, %0, %0, %0")
// the accumulate loop's own mix: per product column 13 v_mad_u64_u32, then and / shift (the 30-bit carry)
__global__ void k_c_column_mix(u64* out, u32 a, u32 b, unsigned long long* cyc) {
  u64 acc[4]; u32 lo[4]; for (int i = 0; i < 4; i++) { acc[i] = threadIdx.x + i; lo[i] = 0; }
  const u32 x = a + threadIdx.x;
  const u64 t0 = memtime();
  for (int it = 0; it < ITERS / 8; it++) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
#pragma unroll
      for (int k = 0; k < 13; k++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(x), "v"(b) : "vcc");
      asm volatile("v_and_b32 %0, %2, %1" : "=v"(lo[i]) : "v"((u32)acc[i]), "s"(0x3fffffffu));
      asm volatile("v_add_u32 %0, %0, %1" : "+v"(lo[i]) : "v"(x));
      asm volatile("v_lshrrev_b64 %0, 30, %0" : "+v"(acc[i]));
    }
  }
  const u64 t1 = memtime();
  if ((threadIdx.x & 63) == 0) atomicAdd(cyc, (unsigned long long)(t1 - t0));
  u64 s = 0; for (int i = 0; i < 4; i++) s += acc[i] + lo[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <class F>
__global__ void k_ffmul(F* out, const F* in, int iters) {
  int tid = blockIdx.x * blockDim.x + threadIdx.x;
  F x = ff_load(in + tid), y = ff_load(in + tid + 1);
  for (int it = 0; it < iters; it++) { F z = ff_mul(x, y); y = x; x = z; }
  ff_store(out + tid, x);
}
template <class F>
__global__ void k_ffadd(F* out, const F* in, int iters) {
  int tid = blockIdx.x * blockDim.x + threadIdx.x;
  F x = ff_load(in + tid), y = ff_load(in + tid + 1);
  for (int it = 0; it < iters; it++) { F z = ff_add(x, y); y = ff_sub(x, z); x = z; }
  ff_store(out + tid, x);
}
#include "../marlin_amd/csrc/fq30.cuh"
// experiment: the same multiplication with every product issued as an inline-asm v_mad_u64_u32 on ONE accumulator per
// column (hipcc otherwise splits the column sums into several chains and adds them up with v_lshl_add_u64)
__device__ __forceinline__ void mad64(u64& acc, u32 a, u32 b) {
  asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b) : "vcc");
}
__device__ __forceinline__ void mad64s(u64& acc, u32 a, u32 k) {
  asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "s"(k) : "vcc");
}
__device__ __forceinline__ Fq30 f30_mul_asm(const Fq30& a, const Fq30& b) {
  constexpr int NL = Fq30::NL;
  using PP = Fq30Params;
  u32 t[2 * NL];
  u64 acc = 0;
#pragma unroll
  for (int k = 0; k < 2 * NL - 1; k++) {
#pragma unroll
    for (int i = (k < NL ? 0 : k - NL + 1); i <= (k < NL ? k : NL - 1); i++) mad64(acc, a.v[i], b.v[k - i]);
    t[k] = (u32)acc & M30;
    acc >>= 30;
  }
  t[2 * NL - 1] = (u32)acc;
  u32 m[NL];
  Fq30 r;
  acc = 0;
#pragma unroll
  for (int k = 0; k < NL; k++) {
    acc += t[k];
#pragma unroll
    for (int i = 0; i < k; i++) mad64s(acc, m[i], PP::P[k - i]);
    m[k] = ((u32)acc * PP::PINV) & M30;
    mad64s(acc, m[k], PP::P[0]);
    acc >>= 30;
  }
#pragma unroll
  for (int k = NL; k < 2 * NL; k++) {
    acc += t[k];
#pragma unroll
    for (int i = k - NL + 1; i < NL; i++) mad64s(acc, m[i], PP::P[k - i]);
    if (k < 2 * NL - 1) { r.v[k - NL] = (u32)acc & M30; acc >>= 30; }
    else r.v[k - NL] = (u32)acc;
  }
  return r;
}
__global__ void k_mul30asm(Fq30* out, const Fq* in, int iters) {
  int tid = blockIdx.x * blockDim.x + threadIdx.x;
  Fq30 x = f30_split(ff_load(in + tid)), y = f30_split(ff_load(in + tid + 1));
  for (int it = 0; it < iters; it++) { Fq30 z = f30_mul_asm(x, y); y = x; x = z; }
  out[tid] = x;
}
__global__ void k_mul30(Fq30* out, const Fq* in, int iters) {
  int tid = blockIdx.x * blockDim.x + threadIdx.x;
  Fq30 x = f30_split(ff_load(in + tid)), y = f30_split(ff_load(in + tid + 1));
  for (int it = 0; it < iters; it++) { Fq30 z = f30_mul(x, y); y = x; x = z; }
  out[tid] = x;
}
__global__ void k_addsub30(Fq30* out, const Fq* in, int iters) {
  int tid = blockIdx.x * blockDim.x + threadIdx.x;
  Fq30 x = f30_split(ff_load(in + tid)), y = f30_split(ff_load(in + tid + 1));
  for (int it = 0; it < iters; it++) { Fq30 z = f30_add(x, y); y = f30_sub<8>(x, z); x = f30_mul(z, y); }
  out[tid] = x;
}
// correctness: (a * b) through both representations must agree
__global__ void k_check30(u32* bad, const Fq* in, int n) {
  int tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= n) return;
  Fq a = ff_load(in + tid), b = ff_load(in + tid + 1);
  // inputs are arbitrary 380-bit integers: reduce them once so that both are < p
  a = ff_mul(a, Fq::one()); b = ff_mul(b, Fq::one());
  Fq want = ff_mul(a, b);
  Fq30 a30 = f30_from_fq(a), b30 = f30_from_fq(b);
  Fq30 c30 = f30_mul(a30, b30);
  // exercise lazy add/sub:  (c + a) - a, and 8p-offset subtraction
  Fq30 d30 = f30_sub<2>(f30_add(c30, a30), a30);
  Fq got = f30_to_fq(d30);
  // also (a - b) * (a + b) == a^2 - b^2
  Fq30 e30 = f30_mul(f30_sub<2>(a30, b30), f30_add(a30, b30));
  Fq30 f30v = f30_sub<2>(f30_sqr(a30), f30_sqr(b30));
  Fq ge = f30_to_fq(e30), gf = f30_to_fq(f30v);
  Fq z = ff_sub(a, a);
  bool ok = true;
  for (int i = 0; i < Fq::N; i++) ok = ok && got.v[i] == want.v[i] && ge.v[i] == gf.v[i];
  ok = ok && f30_is_zero(f30_sub<2>(a30, a30)) && f30_is_zero(f30_sub<8>(f30_add(a30, a30), f30_dbl(a30)));
  (void)z;
  if (!ok) atomicAdd(bad, 1u);
}
__global__ void k_madd(G1Xyzz* out, const Fq* in, int iters) {
  int tid = blockIdx.x * blockDim.x + threadIdx.x;
  G1Xyzz acc; acc.x = ff_load(in + tid); acc.y = ff_load(in + tid + 1); acc.zz = ff_load(in + tid + 2); acc.zzz = ff_load(in + tid + 3);
  Fq px = ff_load(in + tid + 4), py = ff_load(in + tid + 5);
  for (int it = 0; it < iters; it++) { g1_madd(acc, px, py); px = ff_add(px, py); }
  g1_store_xyzz(out + tid, acc);
}

template <class K, class... A>
double timeit(K k, dim3 g, dim3 b, int reps, A... args) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k, g, b, 0, 0, args...);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k, g, b, 0, 0, args...);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device %s %s CUs=%d clock=%d kHz\n", p.name, p.gcnArchName, p.multiProcessorCount, p.clockRate);
  const int blocks = p.multiProcessorCount * 8, threads = 256;
  const size_t nthreads = (size_t)blocks * threads;
  u64* out; CK(hipMalloc(&out, nthreads * 8 + 4096));
  double lanes_ops;
  double ms;
  ms = timeit(k_mad64, dim3(blocks), dim3(threads), 5, out, 3u, 5u); lanes_ops = (double)nthreads * ITERS * 8;
  printf("v_mad_u64_u32      : %8.3f ms  %8.2f Tops/s (lane-ops)\n", ms, lanes_ops / ms / 1e9);
  ms = timeit(k_mullo, dim3(blocks), dim3(threads), 5, out, 3u, 5u);
  printf("v_mul_lo_u32       : %8.3f ms  %8.2f Tops/s\n", ms, lanes_ops / ms / 1e9);
  ms = timeit(k_mulhi, dim3(blocks), dim3(threads), 5, out, 3u, 5u);
  printf("v_mul_hi_u32       : %8.3f ms  %8.2f Tops/s\n", ms, lanes_ops / ms / 1e9);
  ms = timeit(k_addc, dim3(blocks), dim3(threads), 5, out, 3u, 5u);
  printf("v_addc_co_u32      : %8.3f ms  %8.2f Tops/s\n", ms, lanes_ops / ms / 1e9);
  ms = timeit(k_mad24, dim3(blocks), dim3(threads), 5, out, 3u, 5u);
  printf("v_mad_u32_u24      : %8.3f ms  %8.2f Tops/s\n", ms, lanes_ops / ms / 1e9);
  ms = timeit(k_madpair, dim3(blocks), dim3(threads), 5, out, 3u, 5u);
  printf("mad+addc pair      : %8.3f ms  %8.2f Tpairs/s\n", ms, lanes_ops / ms / 1e9);


  // ---- controls with in-kernel cycle counts ------------------------------------------------------------------
  {
    unsigned long long* dcyc; CK(hipMalloc(&dcyc, 8));
    const double nwaves = (double)nthreads / 64, waves_per_simd = (double)blocks / p.multiProcessorCount * threads / 64 / 4;
    printf("controls: %d blocks x %d threads = %.0f waves per SIMD; cyc = shader cycles per wave64 instruction per SIMD (s_memtime)\n",
           blocks, threads, waves_per_simd);
#define RUN_CTRL(K, LABEL, NINSTR)                                                                                          \
    {                                                                                                                       \
      CK(hipMemset(dcyc, 0, 8));                                                                                            \
      hipLaunchKernelGGL(K, dim3(blocks), dim3(threads), 0, 0, out, 3u, 5u, dcyc); CK(hipDeviceSynchronize());             \
      CK(hipMemset(dcyc, 0, 8));                                                                                            \
      ms = timeit(K, dim3(blocks), dim3(threads), 5, out, 3u, 5u, dcyc);                                                    \
      unsigned long long hc = 0; CK(hipMemcpy(&hc, dcyc, 8, hipMemcpyDeviceToHost));                                        \
      const double instr = (double)(NINSTR);                                                                                \
      printf("%-22s: %8.3f ms  %7.2f T lane-ops/s  %5.2f cyc/instr/SIMD  (eff. clock %.2f GHz)\n", LABEL, ms,               \
             (double)nthreads * instr / ms / 1e9, (double)hc / 6.0 / nwaves / instr / waves_per_simd,                      \
             ((double)hc / 6.0 / nwaves) / (ms * 1e6));                                                                     \
    }
    const double N8 = (double)ITERS * 8;
    RUN_CTRL(k_c_fma_f32, "v_fma_f32", N8)
    RUN_CTRL(k_c_pk_fma_f32, "v_pk_fma_f32", N8)
    RUN_CTRL(k_c_fma_f64, "v_fma_f64", N8)
    RUN_CTRL(k_c_add_u32, "v_add_u32", N8)
    RUN_CTRL(k_c_and_b32, "v_and_b32", N8)
    RUN_CTRL(k_c_lshrrev_b32, "v_lshrrev_b32", N8)
    RUN_CTRL(k_c_and_or_b32, "v_and_or_b32", N8)
    RUN_CTRL(k_c_add3_u32, "v_add3_u32", N8)
    RUN_CTRL(k_c_add_co_u32, "v_add_co_u32", N8)
    RUN_CTRL(k_c_addc_co_u32, "v_addc_co_u32", N8)
    RUN_CTRL(k_c_mad_u32_u24, "v_mad_u32_u24", N8)
    RUN_CTRL(k_c_mul_lo_u32, "v_mul_lo_u32", N8)
    RUN_CTRL(k_c_mad_u64_u32, "v_mad_u64_u32", N8)
    RUN_CTRL(k_c_lshrrev_b64, "v_lshrrev_b64", N8)
    RUN_CTRL(k_c_lshl_add_u64, "v_lshl_add_u64", N8)
    RUN_CTRL(k_c_column_mix, "13 mad + and + add + shr", (double)(ITERS / 8) * 4 * 16)
  }

  // field ops
  std::vector<u32> h((nthreads + 8) * 12);
  for (size_t i = 0; i < h.size(); i++) h[i] = (u32)(i * 2654435761u) & 0x0fffffffu;
  void *din, *dout; CK(hipMalloc(&din, h.size() * 4)); CK(hipMalloc(&dout, (nthreads + 8) * 192));
  CK(hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  int it = 512;
  ms = timeit(k_ffmul<Fr>, dim3(blocks), dim3(threads), 3, (Fr*)dout, (const Fr*)din, it);
  printf("Fr mont mul (8 limb) : %8.3f ms  %8.2f Gmul/s\n", ms, (double)nthreads * it / ms / 1e6);
  ms = timeit(k_ffmul<Fq>, dim3(blocks), dim3(threads), 3, (Fq*)dout, (const Fq*)din, it);
  printf("Fq mont mul (12 limb): %8.3f ms  %8.2f Gmul/s\n", ms, (double)nthreads * it / ms / 1e6);
  ms = timeit(k_ffadd<Fq>, dim3(blocks), dim3(threads), 3, (Fq*)dout, (const Fq*)din, it);
  printf("Fq add+sub           : %8.3f ms  %8.2f Gpair/s\n", ms, (double)nthreads * it / ms / 1e6);
  ms = timeit(k_mul30, dim3(blocks), dim3(threads), 3, (Fq30*)dout, (const Fq*)din, it);
  printf("Fq 13x30-bit lazy mul: %8.3f ms  %8.2f Gmul/s\n", ms, (double)nthreads * it / ms / 1e6);
  ms = timeit(k_mul30asm, dim3(blocks), dim3(threads), 3, (Fq30*)dout, (const Fq*)din, it);
  printf("Fq 13x30 mul, asm mads: %8.3f ms  %8.2f Gmul/s\n", ms, (double)nthreads * it / ms / 1e6);
  ms = timeit(k_addsub30, dim3(blocks), dim3(threads), 3, (Fq30*)dout, (const Fq*)din, it);
  printf("Fq30 add+sub+mul     : %8.3f ms  %8.2f Gtriple/s\n", ms, (double)nthreads * it / ms / 1e6);
  {
    u32* dbad; CK(hipMalloc(&dbad, 4)); CK(hipMemset(dbad, 0, 4));
    hipLaunchKernelGGL(k_check30, dim3(blocks), dim3(threads), 0, 0, dbad, (const Fq*)din, nthreads);
    u32 hb = 1; CK(hipMemcpy(&hb, dbad, 4, hipMemcpyDeviceToHost));
    printf("Fq30 vs Fq check     : %u mismatches of %d\n", hb, nthreads);
  }
  it = 64;
  ms = timeit(k_madd, dim3(blocks), dim3(128), 3, (G1Xyzz*)dout, (const Fq*)din, it);
  printf("G1 XYZZ madd         : %8.3f ms  %8.2f Gadd/s\n", ms, (double)blocks * 128 * it / ms / 1e6);
  return 0;
}
