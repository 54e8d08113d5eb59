// Pippenger multi-scalar multiplication over G2 and the fixed-base powers of KZG10::setup's G2 side.
//
// Replaces ark_ec::msm::VariableBaseMSM::multi_scalar_mul over G2Affine and the G2 half of kzg10::setup
// (`powers_of_h` / `neg_powers_of_h`, reached from /root/reference src/lib.rs:79-96 through PC::setup) [ark-* 0.3,
// third-party, UPSTREAM-RECALLED].  Off the prover's hot path (the prover multiplies G1 points only), so it reuses the
// group-agnostic stages of msm.cuh as they are -- signed-digit recoding, per-tile histograms, scans, scatter into
// per-(window, bucket) lists -- and adds only what depends on the group: thread-per-bucket accumulation with the
// complete mixed addition, the segmented bucket reduction, and a device-side combination of the windows down to ONE
// affine point (so the host needs no Fq2 arithmetic at all).
#pragma once
#include "msm.cuh"
#include "g2.cuh"

namespace msmg2 {

__global__ __launch_bounds__(128) void accum_kernel(const G2Affine* __restrict__ bases, const u32* __restrict__ sorted, u64 n,
                                                    const u32* __restrict__ base, const u32* __restrict__ tot,
                                                    G2Xyzz* __restrict__ buckets, u32 nb, u64 WB) {
  const u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= WB) return;
  const u32 w = (u32)(gid / nb);
  const u32* lst = sorted + (u64)w * n + base[gid];
  const u32 cnt = tot[gid];
  G2Xyzz acc = G2Xyzz::identity();
  for (u32 k = 0; k < cnt; k++) {
    const u32 e = lst[k];
    G2Affine p = g2_load_affine(bases + (e & 0x7fffffffu));
    if (e & 0x80000000u) p.y = f2_neg(p.y);
    g2_madd(acc, p.x, p.y);
  }
  g2_store_xyzz(buckets + gid, acc);
}

// thread per (window, segment of `seg` buckets): sum_b (b + 1) B_b over the segment (running sums + offset by
// double-and-add), as msm::reduce1_kernel
__global__ __launch_bounds__(64) void reduce1_kernel(const G2Xyzz* __restrict__ buckets, G2Xyzz* __restrict__ segsum, u32 nb,
                                                     u32 nseg, u32 W, u32 seg) {
  const u32 gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= W * nseg) return;
  const u32 w = gid / nseg, s = gid % nseg;
  const u32 lo = s * seg;
  u32 hi = lo + seg; if (hi > nb) hi = nb;
  G2Xyzz running = G2Xyzz::identity(), acc = G2Xyzz::identity();
  const G2Xyzz* B = buckets + (u64)w * nb;
  for (u32 b = hi; b-- > lo;) {
    const G2Xyzz t = g2_load_xyzz(B + b);
    g2_add(running, t);
    g2_add(acc, running);
  }
  if (lo) {
    G2Xyzz m = G2Xyzz::identity();
    for (int bit = 31 - __clz(lo); bit >= 0; bit--) {
      g2_dbl(m);
      if ((lo >> bit) & 1) g2_add(m, running);
    }
    g2_add(acc, m);
  }
  g2_store_xyzz(segsum + gid, acc);
}

// block per window: tree sum of its nseg segment results
__global__ __launch_bounds__(64) void reduce2_kernel(const G2Xyzz* __restrict__ segsum, G2Xyzz* __restrict__ winsum, u32 nseg) {
  __shared__ G2Xyzz sh[64];
  const u32 w = blockIdx.x;
  G2Xyzz acc = G2Xyzz::identity();
  for (u32 s = threadIdx.x; s < nseg; s += 64) {
    const G2Xyzz t = g2_load_xyzz(segsum + (u64)w * nseg + s);
    g2_add(acc, t);
  }
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (u32 off = 32; off > 0; off >>= 1) {
    if (threadIdx.x < off) {
      G2Xyzz a = sh[threadIdx.x];
      const G2Xyzz b = sh[threadIdx.x + off];
      g2_add(a, b);
      sh[threadIdx.x] = a;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) g2_store_xyzz(winsum + w, sh[0]);
}

// one thread: result = sum_w 2^start_w S_w (Horner from the top window down), normalised to affine.
// out: x.c0 | x.c1 | y.c0 | y.c1 (Montgomery Fq words) followed by one u32 infinity flag
__global__ void combine_kernel(const G2Xyzz* __restrict__ winsum, u32 W, msm::Windows win, u32* __restrict__ out) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  G2Xyzz acc = G2Xyzz::identity();
  for (int w = (int)W - 1; w >= 0; w--) {
    const G2Xyzz s = g2_load_xyzz(winsum + w);
    g2_add(acc, s);
    if (w > 0) for (u32 k = 0; k < win.bits[w - 1]; k++) g2_dbl(acc);
  }
  G2Affine a;
  u32 inf = 0;
  if (acc.is_identity()) { a.x = Fq2::zero(); a.y = Fq2::zero(); inf = 1; }
  else a = g2_to_affine(acc);
  f2_store(reinterpret_cast<Fq2*>(out), a.x);
  f2_store(reinterpret_cast<Fq2*>(out) + 1, a.y);
  out[4 * Fq::N] = inf;
}

// bases[i] = [scale tau^(first + i)] H (H affine, any point of G2): KZG10::setup's powers_of_h / neg_powers_of_h for a
// known-tau (test / bench) SRS.  One thread per power: scalar by square-and-multiply, point by double-and-add.
__global__ __launch_bounds__(64) void powers_kernel(G2Affine* __restrict__ bases, G2Affine H, Fr tau, Fr scale, u64 first, u64 n,
                                                    u32* __restrict__ zero_seen) {
  const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Fr s = ff_from_mont(ff_mul(scale, ff_pow(tau, first + i)));     // canonical integer
  G2Xyzz acc = G2Xyzz::identity();
  for (int limb = Fr::N - 1; limb >= 0; limb--)
    for (int b = 31; b >= 0; b--) {
      g2_dbl(acc);
      if ((s.v[limb] >> b) & 1u) g2_madd(acc, H.x, H.y);
    }
  if (acc.is_identity()) { atomicAdd(zero_seen, 1u); f2_store(&bases[i].x, Fq2::zero()); f2_store(&bases[i].y, Fq2::zero()); return; }
  const G2Affine a = g2_to_affine(acc);
  f2_store(&bases[i].x, a.x); f2_store(&bases[i].y, a.y);
}

__global__ __launch_bounds__(128) void check_kernel(const G2Affine* __restrict__ pts, u64 n, u32* __restrict__ bad) {
  const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (!g2_on_curve(g2_load_affine(pts + i))) atomicAdd(bad, 1u);
}

}  // namespace msmg2
