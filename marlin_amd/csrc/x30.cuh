// Points on the 30-bit-limb field arithmetic of fq30.cuh: the affine table point of the fixed-base MSM and the XYZZ point its
// buckets, row / column sums and planes are (msm_fb.cuh).  Device-inline code only -- no kernel lives here, so the file can be
// included by more than one translation unit (testhooks.hip holds the self-test kernel of this arithmetic).
#pragma once
#include "g1.cuh"
#include "fq30.cuh"

namespace msmfb {

// Table point: affine, coordinates as 30-bit-limb Montgomery residues (fq30.cuh), each coordinate padded to a
// multiple of four words (BLS12-381: 2 x 64 B, one cache line per coordinate)
constexpr int LIMB_SLOTS = (Fq30::NL + 3) & ~3;
struct G1Aff30 { u32 x[LIMB_SLOTS]; u32 y[LIMB_SLOTS]; };

__device__ __forceinline__ Fq30 load30(const u32* __restrict__ p) {
  u32 w[LIMB_SLOTS];
#pragma unroll
  for (int i = 0; i < LIMB_SLOTS; i += 4) {
    uint4 q = *reinterpret_cast<const uint4*>(p + i);
    w[i] = q.x; w[i + 1] = q.y; w[i + 2] = q.z; w[i + 3] = q.w;
  }
  Fq30 r;
#pragma unroll
  for (int i = 0; i < Fq30::NL; i++) r.v[i] = w[i];
  return r;
}
__device__ __forceinline__ void store30(u32* p, const Fq30& a) {
#pragma unroll
  for (int i = 0; i < LIMB_SLOTS; i += 4) {
    uint4 q;
    q.x = i < Fq30::NL ? a.v[i] : 0; q.y = i + 1 < Fq30::NL ? a.v[i + 1] : 0;
    q.z = i + 2 < Fq30::NL ? a.v[i + 2] : 0; q.w = i + 3 < Fq30::NL ? a.v[i + 3] : 0;
    *reinterpret_cast<uint4*>(p + i) = q;
  }
}

// ---- XYZZ points on 30-bit limbs (buckets between accumulate and the bucket reduction) -------------------------
// Bounds kept by every operation below (units of p): X <= 6.2, Y <= 3.2, ZZ, ZZZ <= 1.1; the identity is ZZ = 0
// exactly (a ZZ computed by the formulas is a product of non-zero residues).
struct X30 { Fq30 x, y, zz, zzz; };
struct G1Xyzz30 { u32 c[4][LIMB_SLOTS]; };

__device__ __forceinline__ bool x30_is_identity(const X30& a) {
  u32 o = 0;
#pragma unroll
  for (int i = 0; i < Fq30::NL; i++) o |= a.zz.v[i];
  return o == 0;
}
__device__ __forceinline__ X30 x30_identity() {
  X30 r;
#pragma unroll
  for (int i = 0; i < Fq30::NL; i++) { r.x.v[i] = 0; r.y.v[i] = 0; r.zz.v[i] = 0; r.zzz.v[i] = 0; }
  return r;
}
__device__ __forceinline__ X30 x30_load(const G1Xyzz30* p) {
  X30 r;
  r.x = load30(p->c[0]); r.y = load30(p->c[1]); r.zz = load30(p->c[2]); r.zzz = load30(p->c[3]);
  return r;
}
__device__ __forceinline__ void x30_store(G1Xyzz30* p, const X30& a) {
  store30(p->c[0], a.x); store30(p->c[1], a.y); store30(p->c[2], a.zz); store30(p->c[3], a.zzz);
}
__device__ __forceinline__ G1Xyzz x30_to_std(const X30& a) {
  G1Xyzz r;
  if (x30_is_identity(a)) return G1Xyzz::identity();
  r.x = f30_to_fq(a.x); r.y = f30_to_fq(a.y); r.zz = f30_to_fq(a.zz); r.zzz = f30_to_fq(a.zzz);
  return r;
}
__device__ __forceinline__ X30 x30_from_std(const G1Xyzz& a) {
  X30 r;
  r.x = f30_from_fq(a.x); r.y = f30_from_fq(a.y); r.zz = f30_from_fq(a.zz); r.zzz = f30_from_fq(a.zzz);
  return r;
}
// acc += b   [EFD add-2008-s]  12M + 2S
__device__ __forceinline__ void x30_add_inl(X30& acc, const X30& b);
__device__ __noinline__ void x30_dbl(X30& a);
__device__ __noinline__ void x30_add(X30& acc, const X30& b) { x30_add_inl(acc, b); }
__device__ __forceinline__ void x30_add_inl(X30& acc, const X30& b) {
  if (x30_is_identity(b)) return;
  if (x30_is_identity(acc)) { acc = b; return; }
  const Fq30 U1 = f30_mul(acc.x, b.zz);
  const Fq30 S1 = f30_mul(acc.y, b.zzz);
  const Fq30 P = f30_sub<2>(f30_mul(b.x, acc.zz), U1);
  const Fq30 R = f30_sub<2>(f30_mul(b.y, acc.zzz), S1);
  if (__builtin_expect(f30_is_zero(P), 0)) {            // equal x: the same point (doubling) or opposite points (see x30_add_ilp_inl)
    if (f30_is_zero(R)) x30_dbl(acc); else acc = x30_identity();
    return;
  }
  Fq30 PP = f30_sqr(P);
  const Fq30 Q = f30_mul(U1, PP);
  acc.zz = f30_mul(f30_mul(acc.zz, b.zz), PP);
  PP = f30_mul(P, PP);                                  // PPP
  acc.zzz = f30_mul(f30_mul(acc.zzz, b.zzz), PP);
  acc.x = f30_sub2<3>(f30_sub<2>(f30_sqr(R), PP), Q);
  acc.y = f30_sub<2>(f30_mul(R, f30_sub<8>(Q, acc.x)), f30_mul(S1, PP));
}
// a = 2 a   [EFD dbl-2008-s-1, a = 0]; a point of G1 has odd order, so 2 a is never the identity
__device__ __noinline__ void x30_dbl(X30& a) {
  if (x30_is_identity(a)) return;
  const Fq30 U = f30_dbl(a.y);
  const Fq30 V = f30_sqr(U);
  const Fq30 W = f30_mul(U, V);
  const Fq30 S = f30_mul(a.x, V);
  const Fq30 XX = f30_sqr(a.x);
  const Fq30 M = f30_add(f30_dbl(XX), XX);
  const Fq30 X3 = f30_sub2<3>(f30_sqr(M), S);
  a.y = f30_sub<2>(f30_mul(M, f30_sub<8>(S, X3)), f30_mul(W, a.y));
  a.zz = f30_mul(V, a.zz);
  a.zzz = f30_mul(W, a.zzz);
  a.x = X3;
}

// ---- the same group law with its independent multiplications side by side (fq30.cuh f30_mul_x3 ...) -------------------
// Inside a general addition every field multiplication is one chain of ~400 dependent instructions, but the addition's 14
// multiplications are not all dependent on each other: {U1, S1, ZZ1 ZZ2}, {U2, S2, ZZZ1 ZZZ2}, {P^2, R^2}, {Q, PPP, ZZ3}, {Y3, ZZZ3}
// -- five steps of 2-3 interleaved chains instead of 14 single ones; the doubling's 10 become 4 steps.  Built in round 4 for the
// segment reduction, which ran one wave per SIMD (sort + reduce stages 11.85 -> 11.57 ms per proof at 2^20,
// profiles/r04ef_ab_reduce_ilp_and_called_field_ops.txt); round 5's row / column sums use the same law, and the sweep of their
// launch shape shows what the interleaving buys: ONE such wave already saturates its SIMD's VALU (one and two resident waves run
// at the same rate, profiles/r05c_sweep_rsum_threads.txt, r05w_sq_counters_rsum_kernel_wave_cycle_breakdown.json).  (CALLING the
// multi-chain operations instead of inlining them costs 3.3 ms per proof: the operands travel through scratch.)  Same values mod p
// (Y3 is taken as ONE reduction of R (Q - X3) + (2p - S1) PPP, so its lazy representative differs from x30_add's: compared through
// the canonical form by mh_selftest_fq30; bounds: X <= 6.2, Y <= 1.2, ZZ, ZZZ <= 1.1).
__device__ __noinline__ void x30_dbl_ilp(X30& a);
__device__ __forceinline__ void x30_add_ilp_inl(X30& acc, const X30& b) {
  if (x30_is_identity(b)) return;
  if (x30_is_identity(acc)) { acc = b; return; }
  Fq30 U1, S1, U2, S2, ZZ12, ZZZ12;
  f30_mul_x3(U1, acc.x, b.zz, S1, acc.y, b.zzz, ZZ12, acc.zz, b.zz);
  f30_mul_x3(U2, b.x, acc.zz, S2, b.y, acc.zzz, ZZZ12, acc.zzz, b.zzz);
  const Fq30 P = f30_sub<2>(U2, U1);
  const Fq30 R = f30_sub<2>(S2, S1);
  // Equal x is NOT rare in the bucket reduction: with an empty bucket right after a segment's first non-empty one the running sum
  // and the accumulator are the same point (acc = running = B), and acc += running is a doubling -- at 2^16 a quarter of the
  // waves of an H-sized job meet one.  Round 3 sent these through the complete law in the 32-bit representation (four
  // conversions each way around g1_add, ~5 additions' time, with the whole wave waiting); here the two cases are told apart
  // by R -- same point: the 30-bit doubling; opposite points: the identity.
  if (__builtin_expect(f30_is_zero(P), 0)) {
    if (f30_is_zero(R)) x30_dbl_ilp(acc); else acc = x30_identity();
    return;
  }
  Fq30 PP, RR;
  f30_sqr_x2(PP, P, RR, R);
  Fq30 Q, PPP;
  f30_mul_x3(Q, U1, PP, PPP, P, PP, acc.zz, ZZ12, PP);
  acc.x = f30_sub2<3>(f30_sub<2>(RR, PPP), Q);
  Fq30 zero;
#pragma unroll
  for (int i = 0; i < Fq30::NL; i++) zero.v[i] = 0;
  // R <= 4 p, Q - X3 + 8p <= 10 p, (2p - S1) PPP <= 4 p^2: 44 p^2 < 64 p^2
  f30_mul2_mul(acc.y, R, f30_sub<8>(Q, acc.x), f30_sub<2>(zero, S1), PPP, acc.zzz, ZZZ12, PPP);
}
__device__ __noinline__ void x30_add_ilp(X30& acc, const X30& b) { x30_add_ilp_inl(acc, b); }
__device__ __noinline__ void x30_dbl_ilp(X30& a) {
  if (x30_is_identity(a)) return;
  const Fq30 U = f30_dbl(a.y);
  Fq30 V, XX;
  f30_sqr_x2(V, U, XX, a.x);
  const Fq30 M = f30_add(f30_dbl(XX), XX);
  Fq30 W, S, MM;
  f30_mul_x3(W, U, V, S, a.x, V, a.zz, V, a.zz);
  f30_sqr_mul(MM, M, a.zzz, W, a.zzz);
  const Fq30 X3 = f30_sub2<3>(MM, S);
  Fq30 zero;
#pragma unroll
  for (int i = 0; i < Fq30::NL; i++) zero.v[i] = 0;
  // M <= 3.7 p, S - X3 + 8p <= 10 p, (2p - W) Y <= 6.4 p^2: 44 p^2 < 64 p^2
  a.y = f30_mul2(M, f30_sub<8>(S, X3), f30_sub<2>(zero, W), a.y);
  a.x = X3;
}

}  // namespace msmfb
