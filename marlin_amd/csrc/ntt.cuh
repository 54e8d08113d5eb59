// Radix-2^B Stockham NTT over the scalar field Fr (BLS12-381, or BN254 with -DMH_CURVE_BN254) for gfx950.
//
// Computes what the reference obtains from ark-poly 0.3
// `GeneralEvaluationDomain::{fft, ifft}` (call sites: /root/reference
// src/ahp/prover.rs:326,350-351,359,365,427,488,532-535,545,655,681 and the
// FFT-based `DensePolynomial` products at prover.rs:352,360,366,467,685;
// SURVEY.md Appendix A): natural order in, natural order out, omega = the
// 2-adic root squared down to order n, inverse includes the n^-1 factor.
//
// Design (not a translation of arkworks' DIF + bit-reverse loops):
//  * ceil(log n / 8) out-of-place passes; each pass performs B <= 8 radix-2 DIT
//    stages on an R x C tile (R = 2^B rows gathered at stride n/R, C = 4
//    adjacent columns = 128 contiguous bytes per row) held in LDS, so HBM is
//    read once and written once per pass, always in >= 128-B runs.  (C = 8 was
//    measured 7-12 % slower: its 69 KB tile leaves 2 blocks per CU to cover the
//    per-stage barriers, the 35 KB tile leaves 4.)
//  * Stockham indexing makes the output of the last pass land in natural order
//    with no separate bit-reversal pass.
//  * One twiddle table for all sizes: tw[2^(l-1) + e] = omega_{2^l}^e.  The
//    inverse transform reuses it (forward transform + index negation + n^-1
//    fused into the last pass's store).
#pragma once
#include "ff.cuh"

namespace ntt {

constexpr int MAX_B = 8;          // stages per pass
constexpr int MAX_LOGC = 2;       // 4 columns per tile (128-B runs; 35 KB of LDS -> 4 blocks per CU)
constexpr int THREADS = 256;

// LDS row stride in uint4 units: 2*C data + 1 pad (breaks the 256-B power-of-two
// stride so the transposed store-phase reads are conflict-free for ds_read_b128).
__device__ __forceinline__ int lds_index(int row, int c, int logc) { return row * ((2 << logc) + 1) + 2 * c; }

__device__ __forceinline__ Fr lds_load(const uint4* s, int idx) {
  uint4 a = s[idx], b = s[idx + 1];
  Fr r;
  r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
  r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
  return r;
}
__device__ __forceinline__ void lds_store(uint4* s, int idx, const Fr& r) {
  s[idx] = make_uint4(r.v[0], r.v[1], r.v[2], r.v[3]);
  s[idx + 1] = make_uint4(r.v[4], r.v[5], r.v[6], r.v[7]);
}

// tw table builder: entry i in [1, 2^max_log) : level l = floor(log2 i)+1, e = i - 2^(l-1)
// value = omega_{2^l}^e, omega_{2^l} = root32 ^ (2^(32-l)).
__global__ void build_twiddles(Fr* tw, u32 max_log, Fr root32 /* the 2^FR_TWO_ADICITY-th root */) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  u64 n = 1ull << max_log;
  if (i >= n) return;
  if (i == 0) { ff_store(tw, Fr::one()); return; }
  u32 l = 64 - __clzll(i);            // level: 2^(l-1) <= i < 2^l
  u64 e = i - (1ull << (l - 1));
  // omega_{2^l}^e = root32^(e * 2^(32-l)); e < 2^(l-1) so the exponent < 2^31
  Fr r = ff_pow(root32, e << (FR_TWO_ADICITY - l));
  ff_store(tw + i, r);
}

// One Stockham pass.  x, y: n = 2^log_n elements.  B stages, P = 2^logP = product of
// the previous passes' radices.  flags bit0: this is the last pass of an inverse
// transform (negate index, multiply by ninv).
// in_len: elements of x at index >= in_len are read as zero (a shorter coefficient vector transformed on a larger
// domain needs no padded copy); n for every pass but the first.
template <int LOGC>
__global__ __launch_bounds__(THREADS) void pass_kernel(const Fr* __restrict__ x, Fr* __restrict__ y,
                                                       const Fr* __restrict__ tw, u32 log_n, u32 B, u32 logP,
                                                       u32 flags, Fr ninv, u64 in_len) {
  extern __shared__ __attribute__((aligned(16))) uint4 smem[];
  constexpr int C = 1 << LOGC;
  const u32 R = 1u << B;
  const u64 n = 1ull << log_n;
  const u64 stride = n >> B;                 // n / R
  const u64 P = 1ull << logP;
  const u64 jbase = (u64)blockIdx.x << LOGC; // first column of this tile
  const int tid = threadIdx.x;

  // ---- load: row r (stride n/R), column c -> LDS row bitrev_B(r)
  for (u32 idx = tid; idx < (R << LOGC); idx += THREADS) {
    u32 c = idx & (C - 1);
    u32 r = idx >> LOGC;
    const u64 gi = (jbase + c) + (u64)r * stride;
    Fr v = gi < in_len ? ff_load(x + gi) : Fr::zero();
    u32 rr = B ? (__brev(r) >> (32 - B)) : 0;
    lds_store(smem, lds_index(rr, c, LOGC), v);
  }
  __syncthreads();

  // ---- B radix-2 DIT stages
  for (u32 t = 0; t < B; t++) {
    const u32 m = 1u << t;
    const Fr* twl = tw + (P << t);           // level base: P*m
    for (u32 idx = tid; idx < ((R >> 1) << LOGC); idx += THREADS) {
      u32 c = idx & (C - 1);
      u32 q = idx >> LOGC;
      u32 i = q & (m - 1);
      u32 u = ((q >> t) << (t + 1)) + i;
      u64 k = (jbase + c) & (P - 1);
      Fr w = ff_load(twl + k + ((u64)i << logP));
      int ia = lds_index(u, c, LOGC), ib = lds_index(u + m, c, LOGC);
      Fr a = lds_load(smem, ia);
      Fr b = ff_mul(lds_load(smem, ib), w);
      lds_store(smem, ia, ff_add(a, b));
      lds_store(smem, ib, ff_sub(a, b));
    }
    __syncthreads();
  }

  // ---- store
  const bool inv_last = flags & 1u;
  if (logP == 0) {
    // first pass: column c's outputs are contiguous: y[j*R + r']
    for (u32 idx = tid; idx < (R << LOGC); idx += THREADS) {
      u32 r2 = idx & (R - 1);
      u32 c = idx >> B;
      Fr v = lds_load(smem, lds_index(r2, c, LOGC));
      u64 o = ((jbase + c) << B) + r2;
      if (inv_last) { o = (n - o) & (n - 1); v = ff_mul(v, ninv); }
      ff_store(y + o, v);
    }
  } else {
    for (u32 idx = tid; idx < (R << LOGC); idx += THREADS) {
      u32 c = idx & (C - 1);
      u32 r2 = idx >> LOGC;
      Fr v = lds_load(smem, lds_index(r2, c, LOGC));
      u64 j = jbase + c;
      u64 k = j & (P - 1);
      u64 o = ((j - k) << B) + k + ((u64)r2 << logP);
      if (inv_last) { o = (n - o) & (n - 1); v = ff_mul(v, ninv); }
      ff_store(y + o, v);
    }
  }
}

// ark_poly `distribute_powers(coeffs, g)`: out[i] = in[i] * g^i (coset_fft: before the transform; coset_ifft: after it,
// with g^-1).  A thread owns DP_CH consecutive elements: g^(first) by square-and-multiply, then a running product.
constexpr int DP_CH = 16;
__global__ __launch_bounds__(256) void distribute_powers_kernel(Fr* __restrict__ out, const Fr* __restrict__ in, Fr g, u64 n) {
  const u64 first = ((u64)blockIdx.x * blockDim.x + threadIdx.x) * DP_CH;
  if (first >= n) return;
  Fr p = ff_pow(g, first);
  for (int k = 0; k < DP_CH && first + k < n; k++) {
    ff_store(out + first + k, ff_mul(ff_load(in + first + k), p));
    p = ff_mul(p, g);
  }
}

inline size_t pass_lds_bytes(u32 B, int logc) { return (size_t)(1u << B) * ((2u << logc) + 1) * 16; }

}  // namespace ntt
