// G2 of BLS12-381 / BN254: the sextic twist E'(Fq2): y^2 = x^3 + b', Fq2 = Fq[u] / (u^2 + 1), on device.
//
// Where G2 appears around the reference: `Marlin::universal_setup` -> `PC::setup` (/root/reference src/lib.rs:79-96) builds
// the verifier side of the KZG10 SRS -- h, beta_h and, for SonicKZG10 (benches/bench.rs:81), `neg_powers_of_h`, one
// [-tau^(max_degree - d)]H per enforceable degree bound [ark-poly-commit 0.3 kzg10::setup(produce_g2_powers),
// UPSTREAM-RECALLED; SURVEY.md 8f rank 3] -- and BASELINE.json's north_star names "Pippenger MSM over G1/G2".  The prover
// itself multiplies G1 points only, so this file is deliberately plain: the same XYZZ
// formulas as g1.cuh written once over a field type, with the 32-bit Montgomery arithmetic of ff.cuh underneath, and a
// Pippenger that reuses the group-agnostic sort stages of msm.cuh (msm_g2.cuh).
#pragma once
#include "ff.cuh"

struct Fq2 {
  Fq c0, c1;      // c0 + c1 u
  static __device__ __forceinline__ Fq2 zero() { Fq2 r; r.c0 = Fq::zero(); r.c1 = Fq::zero(); return r; }
  static __device__ __forceinline__ Fq2 one() { Fq2 r; r.c0 = Fq::one(); r.c1 = Fq::zero(); return r; }
  __device__ __forceinline__ bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
};

__device__ __forceinline__ Fq2 f2_add(const Fq2& a, const Fq2& b) { Fq2 r; r.c0 = ff_add(a.c0, b.c0); r.c1 = ff_add(a.c1, b.c1); return r; }
__device__ __forceinline__ Fq2 f2_sub(const Fq2& a, const Fq2& b) { Fq2 r; r.c0 = ff_sub(a.c0, b.c0); r.c1 = ff_sub(a.c1, b.c1); return r; }
__device__ __forceinline__ Fq2 f2_dbl(const Fq2& a) { Fq2 r; r.c0 = ff_dbl(a.c0); r.c1 = ff_dbl(a.c1); return r; }
__device__ __forceinline__ Fq2 f2_neg(const Fq2& a) { Fq2 r; r.c0 = ff_neg(a.c0); r.c1 = ff_neg(a.c1); return r; }
// (a0 + a1 u)(b0 + b1 u) = (a0 b0 - a1 b1) + ((a0 + a1)(b0 + b1) - a0 b0 - a1 b1) u      3 multiplications
__device__ __noinline__ Fq2 f2_mul(const Fq2& a, const Fq2& b) {
  const Fq t0 = ff_mul(a.c0, b.c0), t1 = ff_mul(a.c1, b.c1);
  const Fq t2 = ff_mul(ff_add(a.c0, a.c1), ff_add(b.c0, b.c1));
  Fq2 r;
  r.c0 = ff_sub(t0, t1);
  r.c1 = ff_sub(ff_sub(t2, t0), t1);
  return r;
}
// (a0 + a1 u)^2 = (a0 + a1)(a0 - a1) + 2 a0 a1 u                                        2 multiplications
__device__ __noinline__ Fq2 f2_sqr(const Fq2& a) {
  Fq2 r;
  r.c0 = ff_mul(ff_add(a.c0, a.c1), ff_sub(a.c0, a.c1));
  r.c1 = ff_dbl(ff_mul(a.c0, a.c1));
  return r;
}
// 1 / (a0 + a1 u) = (a0 - a1 u) / (a0^2 + a1^2); 0 -> 0
__device__ __noinline__ Fq2 f2_inv(const Fq2& a) {
  const Fq n = ff_inv(ff_add(ff_sqr(a.c0), ff_sqr(a.c1)));
  Fq2 r;
  r.c0 = ff_mul(a.c0, n);
  r.c1 = ff_neg(ff_mul(a.c1, n));
  return r;
}
__device__ __forceinline__ Fq2 f2_load(const Fq2* p) { Fq2 r; r.c0 = ff_load(&p->c0); r.c1 = ff_load(&p->c1); return r; }
__device__ __forceinline__ void f2_store(Fq2* p, const Fq2& a) { ff_store(&p->c0, a.c0); ff_store(&p->c1, a.c1); }

// b' of the twist in Montgomery form: 4 (1 + u) on BLS12-381; 3 / (9 + u) on BN254
__device__ __forceinline__ Fq2 G2_CURVE_B_MONT() {
  Fq2 r;
#ifdef MH_CURVE_BN254
  // 3 / (9 + u) = 3 (9 - u) / 82: computed from small integers so that no 254-bit constant has to be typed in
  const Fq one = Fq::one();
  Fq nine = ff_dbl(ff_dbl(ff_dbl(one))); nine = ff_add(nine, one);                   // 9
  Fq e82 = ff_add(ff_sqr(nine), one);                                               // 82
  const Fq three = ff_add(ff_dbl(one), one);
  const Fq inv82 = ff_inv(e82);
  r.c0 = ff_mul(ff_mul(three, nine), inv82);
  r.c1 = ff_neg(ff_mul(three, inv82));
#else
  const Fq four = ff_dbl(ff_dbl(Fq::one()));
  r.c0 = four; r.c1 = four;
#endif
  return r;
}

struct G2Affine { Fq2 x, y; };                     // 192 B (BLS12-381), no infinity flag
struct G2Xyzz {                                    // x = X / ZZ, y = Y / ZZZ, ZZ^3 = ZZZ^2; ZZ = 0 is the identity
  Fq2 x, y, zz, zzz;
  static __device__ __forceinline__ G2Xyzz identity() { G2Xyzz r; r.x = Fq2::zero(); r.y = Fq2::zero(); r.zz = Fq2::zero(); r.zzz = Fq2::zero(); return r; }
  __device__ __forceinline__ bool is_identity() const { return zz.is_zero(); }
};
__device__ __forceinline__ G2Affine g2_load_affine(const G2Affine* p) { G2Affine r; r.x = f2_load(&p->x); r.y = f2_load(&p->y); return r; }
__device__ __forceinline__ G2Xyzz g2_load_xyzz(const G2Xyzz* p) { G2Xyzz r; r.x = f2_load(&p->x); r.y = f2_load(&p->y); r.zz = f2_load(&p->zz); r.zzz = f2_load(&p->zzz); return r; }
__device__ __forceinline__ void g2_store_xyzz(G2Xyzz* p, const G2Xyzz& a) { f2_store(&p->x, a.x); f2_store(&p->y, a.y); f2_store(&p->zz, a.zz); f2_store(&p->zzz, a.zzz); }

// 2 (affine)   [EFD mdbl-2008-s-1, a = 0]
__device__ __noinline__ void g2_dbl_affine(G2Xyzz& r, const Fq2& x1, const Fq2& y1) {
  const Fq2 U = f2_dbl(y1), V = f2_sqr(U), W = f2_mul(U, V), S = f2_mul(x1, V), xx = f2_sqr(x1);
  const Fq2 M = f2_add(f2_dbl(xx), xx);
  const Fq2 X3 = f2_sub(f2_sqr(M), f2_dbl(S));
  r.y = f2_sub(f2_mul(M, f2_sub(S, X3)), f2_mul(W, y1));
  r.x = X3; r.zz = V; r.zzz = W;
}
// a = 2 a   [EFD dbl-2008-s-1, a = 0]
__device__ __noinline__ void g2_dbl(G2Xyzz& a) {
  if (a.is_identity()) return;
  const Fq2 U = f2_dbl(a.y), V = f2_sqr(U), W = f2_mul(U, V), S = f2_mul(a.x, V), xx = f2_sqr(a.x);
  const Fq2 M = f2_add(f2_dbl(xx), xx);
  const Fq2 X3 = f2_sub(f2_sqr(M), f2_dbl(S));
  a.y = f2_sub(f2_mul(M, f2_sub(S, X3)), f2_mul(W, a.y));
  a.zz = f2_mul(V, a.zz);
  a.zzz = f2_mul(W, a.zzz);
  a.x = X3;
}
// acc += (x2, y2) affine, complete   [EFD madd-2008-s]
__device__ __noinline__ void g2_madd(G2Xyzz& acc, const Fq2& x2, const Fq2& y2) {
  if (acc.is_identity()) { acc.x = x2; acc.y = y2; acc.zz = Fq2::one(); acc.zzz = Fq2::one(); return; }
  const Fq2 P = f2_sub(f2_mul(x2, acc.zz), acc.x);
  const Fq2 R = f2_sub(f2_mul(y2, acc.zzz), acc.y);
  if (P.is_zero()) {
    if (R.is_zero()) g2_dbl_affine(acc, x2, y2);
    else acc = G2Xyzz::identity();
    return;
  }
  const Fq2 PP = f2_sqr(P), PPP = f2_mul(P, PP), Q = f2_mul(acc.x, PP);
  const Fq2 X3 = f2_sub(f2_sub(f2_sqr(R), PPP), f2_dbl(Q));
  acc.y = f2_sub(f2_mul(R, f2_sub(Q, X3)), f2_mul(acc.y, PPP));
  acc.zz = f2_mul(acc.zz, PP);
  acc.zzz = f2_mul(acc.zzz, PPP);
  acc.x = X3;
}
// acc += b, complete   [EFD add-2008-s]
__device__ __noinline__ void g2_add(G2Xyzz& acc, const G2Xyzz& b) {
  if (b.is_identity()) return;
  if (acc.is_identity()) { acc = b; return; }
  const Fq2 U1 = f2_mul(acc.x, b.zz), U2 = f2_mul(b.x, acc.zz);
  const Fq2 S1 = f2_mul(acc.y, b.zzz), S2 = f2_mul(b.y, acc.zzz);
  const Fq2 P = f2_sub(U2, U1), R = f2_sub(S2, S1);
  if (P.is_zero()) {
    if (R.is_zero()) g2_dbl(acc);
    else acc = G2Xyzz::identity();
    return;
  }
  const Fq2 PP = f2_sqr(P), PPP = f2_mul(P, PP), Q = f2_mul(U1, PP);
  const Fq2 X3 = f2_sub(f2_sub(f2_sqr(R), PPP), f2_dbl(Q));
  acc.y = f2_sub(f2_mul(R, f2_sub(Q, X3)), f2_mul(S1, PPP));
  acc.zz = f2_mul(f2_mul(acc.zz, b.zz), PP);
  acc.zzz = f2_mul(f2_mul(acc.zzz, b.zzz), PPP);
  acc.x = X3;
}
// affine (x, y) of a non-identity point: 1 / ZZ = ZZ^2 / ZZZ^2
__device__ __forceinline__ G2Affine g2_to_affine(const G2Xyzz& a) {
  const Fq2 iz = f2_inv(a.zzz);
  const Fq2 izz = f2_mul(f2_sqr(a.zz), f2_sqr(iz));
  G2Affine r;
  r.x = f2_mul(a.x, izz);
  r.y = f2_mul(a.y, iz);
  return r;
}
__device__ __forceinline__ bool g2_on_curve(const G2Affine& p) {
  const Fq2 lhs = f2_sqr(p.y), rhs = f2_add(f2_mul(f2_sqr(p.x), p.x), G2_CURVE_B_MONT());
  const Fq2 d = f2_sub(lhs, rhs);
  return d.is_zero();
}
