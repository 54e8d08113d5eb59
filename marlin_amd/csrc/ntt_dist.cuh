// One transform of n = 2^log_n points spread over G = 2^g ranks (one process per GPU), with ONE all-to-all.
//
// Reference semantics: the same ark-poly Radix2EvaluationDomain::{fft, ifft} as ntt30.cuh (call sites /root/reference
// src/ahp/prover.rs:326,350-351,...; SURVEY.md 8e: "a single transform split across GPUs = 4-step NTT with one all-to-all").
// With n = G m, input index j = j1 + G j2 and output index k = k2 + m k1 (j1, k1 < G; j2, k2 < m):
//     X[k2 + m k1] = sum_{j1} w_G^{j1 k1} * w_n^{j1 k2} * ( sum_{j2} x[j1 + G j2] w_m^{j2 k2} )
// so rank j1, holding the CYCLIC slice x[j1 + G j2] ("C-layout": the layout of coefficient vectors -- independent of how far
// a vector is zero-padded, which is why polynomials of different lengths stay aligned), runs a local m-point transform,
// multiplies by w_n^{j1 k2}, and the ranks exchange so that rank q receives every Y_{j1}[k2] with k2 in its block
// [q m / G, (q + 1) m / G); a G-point transform per k2 finishes.  Rank q then holds X[k] for (k mod m) in its block
// ("M-layout": the layout of evaluation vectors; a point of a subdomain has the same owner in the larger domain), locally
// ordered k1-major: local[k1 (m / G) + t] = X[q m / G + t + m k1].  The inverse runs the steps backwards (M-layout in,
// C-layout out), with n^-1 split into m^-1 (local inverse transform) and G^-1 (folded into the twiddle tables).
// Per rank: 1 / G of the butterflies of the big transform's lower log m stages + 2 multiplications per element (twiddle
// from a hi / lo table pair) + (log G) / 2 per element (the G-point transforms) and 32 n / G^2 bytes to every peer.
#pragma once
#include "ff.cuh"

namespace nttdist {

struct Roots { Fr w[8]; };      // powers 0 .. G / 2 - 1 of the G-th root (inverse root for the inverse transform), G <= 8

// thread per t < chunk: G-point transform across the G chunks of `in` (chunk j1 at in + j1 * chunk), out chunk k1
template <int LOGG>
__global__ __launch_bounds__(256) void gdft_kernel(Fr* __restrict__ out, const Fr* __restrict__ in, u64 chunk, Roots roots) {
  constexpr int G = 1 << LOGG;
  const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= chunk) return;
  Fr a[G];
#pragma unroll
  for (int j = 0; j < G; j++) {
    int r = 0;
#pragma unroll
    for (int b = 0; b < LOGG; b++) r |= ((j >> b) & 1) << (LOGG - 1 - b);
    a[j] = ff_load(in + (u64)r * chunk + t);          // decimation in time: bit-reversed input order
  }
#pragma unroll
  for (int s = 0; s < LOGG; s++) {
    const int half = 1 << s;
#pragma unroll
    for (int i = 0; i < G; i++) {
      if (i & half) continue;
      const int e = (i & (half - 1)) << (LOGG - 1 - s);   // exponent of the G-th root
      const Fr tw = e ? ff_mul(a[i + half], roots.w[e]) : a[i + half];
      a[i + half] = ff_sub(a[i], tw);
      a[i] = ff_add(a[i], tw);
    }
  }
#pragma unroll
  for (int k = 0; k < G; k++) ff_store(out + (u64)k * chunk + t, a[k]);
}

// ---- moving vectors between the replicated and the sliced world (prover.hip: the sliced sections of rounds 2 and 3) -------
// C-layout slice of a replicated coefficient vector: dst[j] = src[r + G j]
__global__ __launch_bounds__(256) void slice_c_kernel(Fr* __restrict__ dst, const Fr* __restrict__ src, u64 nloc, u32 r, u32 G) {
  const u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (j < nloc) ff_store(dst + j, ff_load(src + r + (u64)G * j));
}
// M-layout block of a replicated evaluation vector on a domain of G m points: dst[l] = src[r chunk + (l mod chunk) + m (l / chunk)]
__global__ __launch_bounds__(256) void gather_m_kernel(Fr* __restrict__ dst, const Fr* __restrict__ src, u64 m, u64 chunk, u32 r) {
  const u64 l = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (l < m) ff_store(dst + l, ff_load(src + (u64)r * chunk + (l % chunk) + m * (l / chunk)));
}
// the all-to-all used as an all-gather: the same chunk (a's na elements followed by b's nb) for every peer
__global__ __launch_bounds__(256) void replicate_kernel(Fr* __restrict__ send, const Fr* __restrict__ a, u64 na, const Fr* __restrict__ b,
                                                        u64 nb, u32 G) {
  const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  const u64 ch = na + nb;
  if (i >= ch) return;
  const Fr v = i < na ? ff_load(a + i) : ff_load(b + (i - na));
  for (u32 q = 0; q < G; q++) ff_store(send + (u64)q * ch + i, v);
}
// back to a replicated coefficient vector from the gathered C-layout slices: full[i] = recv[(i mod G) stride + off + i / G]
__global__ __launch_bounds__(256) void unslice_c_kernel(Fr* __restrict__ full, const Fr* __restrict__ recv, u64 len, u32 G, u64 stride, u64 off) {
  const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < len) ff_store(full + i, ff_load(recv + (i % G) * stride + off + i / G));
}

}  // namespace nttdist
