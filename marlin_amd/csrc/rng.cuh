// Device generation of `DensePolynomial::rand(d, zk_rng)` for a ChaCha-based zk_rng:
// /root/reference src/ahp/prover.rs:371 draws 3|H| field elements one after another with
// Fp256::rand (4 x next_u64, clear the top bit, reject if >= r) --
// about 100 MB of ChaCha20 output at 2^20 constraints and the largest serial host loop of the
// prover.  ChaCha is counter mode, so the stream is generated in parallel; the rejection loop
// becomes a stream compaction (flag -> exclusive scan -> scatter), which reproduces exactly the
// sequence the sequential loop would have produced, and reports how many 32-byte candidates
// were consumed so the host RNG continues from the right word.
#pragma once
#include "ff.cuh"

namespace rng {

struct Key { u32 k[8]; };

__device__ __forceinline__ u32 rotl32(u32 x, int n) { return (x << n) | (x >> (32 - n)); }

#define RNG_QR(a, b, c, d)                                                         \
  w[a] += w[b]; w[d] = rotl32(w[d] ^ w[a], 16); w[c] += w[d]; w[b] = rotl32(w[b] ^ w[c], 12); \
  w[a] += w[b]; w[d] = rotl32(w[d] ^ w[a], 8);  w[c] += w[d]; w[b] = rotl32(w[b] ^ w[c], 7);

// candidate j (global, 32 bytes = half a ChaCha block) for j in [c0, c0 + n): cand[j - c0], flag[j - c0]
__global__ __launch_bounds__(256) void candidates_kernel(Fr* __restrict__ cand, u32* __restrict__ flag, Key key,
                                                         int rounds, u64 c0, u64 n) {
  u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;      // one ChaCha block (2 candidates) per thread
  u64 b = (c0 >> 1) + t;
  if (b * 2 >= c0 + n) return;
  u32 st[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key.k[0], key.k[1], key.k[2], key.k[3],
                key.k[4], key.k[5], key.k[6], key.k[7], (u32)b, (u32)(b >> 32), 0u, 0u};
  u32 w[16];
#pragma unroll
  for (int i = 0; i < 16; i++) w[i] = st[i];
  for (int r = 0; r < rounds; r += 2) {
    RNG_QR(0, 4, 8, 12) RNG_QR(1, 5, 9, 13) RNG_QR(2, 6, 10, 14) RNG_QR(3, 7, 11, 15)
    RNG_QR(0, 5, 10, 15) RNG_QR(1, 6, 11, 12) RNG_QR(2, 7, 8, 13) RNG_QR(3, 4, 9, 14)
  }
#pragma unroll
  for (int i = 0; i < 16; i++) w[i] += st[i];
#pragma unroll
  for (int h = 0; h < 2; h++) {
    u64 j = b * 2 + h;
    if (j < c0 || j >= c0 + n) continue;
    Fr x;
#pragma unroll
    for (int i = 0; i < 8; i++) x.v[i] = w[8 * h + i];
    x.v[7] &= FR_SHAVE_MASK_TOP32;                          // clear the REPR_SHAVE_BITS top bits
    // accept iff x < r
    u32 borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { u64 d = (u64)x.v[i] - CurveFrParams::MOD[i] - borrow; borrow = (u32)(d >> 63); }
    ff_store(cand + (j - c0), x);
    flag[j - c0] = borrow;                                  // 1 = accepted
  }
}
#undef RNG_QR

// ---- exclusive scan of u32 flags (3 phases, 1024 elements per block) ---------------------------------
__global__ __launch_bounds__(256) void scan_reduce_kernel(const u32* __restrict__ in, u32* __restrict__ sums, u64 n) {
  __shared__ u32 sh[256];
  u64 base = (u64)blockIdx.x * 1024;
  u32 s = 0;
  for (int k = 0; k < 4; k++) { u64 i = base + threadIdx.x * 4 + k; if (i < n) s += in[i]; }
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) { if ((int)threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off]; __syncthreads(); }
  if (threadIdx.x == 0) sums[blockIdx.x] = sh[0];
}
__global__ __launch_bounds__(1024) void scan_sums_kernel(u32* __restrict__ sums, u64 nblocks, u32* __restrict__ total) {
  __shared__ u32 part[1024];
  u64 per = (nblocks + 1023) / 1024;
  u64 lo = threadIdx.x * per;
  u32 s = 0;
  for (u64 k = 0; k < per; k++) if (lo + k < nblocks) s += sums[lo + k];
  part[threadIdx.x] = s;
  __syncthreads();
  for (u32 off = 1; off < 1024; off <<= 1) {
    u32 v = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  u32 run = part[threadIdx.x] - s;
  for (u64 k = 0; k < per; k++) if (lo + k < nblocks) { u32 v = sums[lo + k]; sums[lo + k] = run; run += v; }
  if (threadIdx.x == 1023) *total = part[1023];
}
// scatter accepted candidates: out[rank] = cand[i] for rank < needed; last_idx = index of the needed-th accept
__global__ __launch_bounds__(256) void scan_scatter_kernel(Fr* __restrict__ out, const Fr* __restrict__ cand,
                                                           const u32* __restrict__ flag, const u32* __restrict__ sums,
                                                           u64 n, u64 needed, u64 out_base, u64* __restrict__ last_idx) {
  __shared__ u32 sh[256];
  u64 base = (u64)blockIdx.x * 1024;
  u32 f[4]; u32 s = 0;
  for (int k = 0; k < 4; k++) { u64 i = base + threadIdx.x * 4 + k; f[k] = i < n ? flag[i] : 0; s += f[k]; }
  sh[threadIdx.x] = s;
  __syncthreads();
  for (u32 off = 1; off < 256; off <<= 1) {
    u32 v = threadIdx.x >= off ? sh[threadIdx.x - off] : 0;
    __syncthreads();
    sh[threadIdx.x] += v;
    __syncthreads();
  }
  u64 rank = (u64)sums[blockIdx.x] + sh[threadIdx.x] - s + out_base;
  for (int k = 0; k < 4; k++) {
    u64 i = base + threadIdx.x * 4 + k;
    if (f[k]) {
      if (rank < needed) { ff_store(out + rank, ff_load(cand + i)); if (rank + 1 == needed) *last_idx = i; }
      rank++;
    }
  }
}

// mask_poly[0] -= sum_{i = 0 .. upper} mask_poly[H * i]      (prover.rs:373-380)
__global__ void mask_fix_kernel(Fr* __restrict__ mask, u64 H, u64 len) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  Fr r0 = Fr::zero();
  for (u64 i = 0; i * H < len; i++) r0 = ff_add(r0, ff_load(mask + i * H));
  ff_store(mask, ff_sub(ff_load(mask), r0));
}

}  // namespace rng
