// BLS12-381 G1 (y^2 = x^3 + 4, a = 0) point arithmetic on device, extended
// Jacobian "XYZZ" accumulators (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2; ZZ = 0 is the
// identity).  Coordinates are Montgomery-form Fq (12 x u32).
//
// Stands in, on device, for ark-ec 0.3's short-Weierstrass `GroupAffine` /
// `GroupProjective` arithmetic used inside `VariableBaseMSM::multi_scalar_mul`
// (third-party).  Any correct group law yields the same group
// element; results are normalised to affine on the host before they are
// serialised or hashed, which is where bit-exactness is defined.
#pragma once
#include "ff.cuh"

struct G1Affine {   // 96 bytes, no infinity flag (SRS points are never the identity)
  Fq x, y;
};

// the curve constant b of y^2 = x^3 + b (4 on BLS12-381, 3 on BN254) in Montgomery form
__device__ __forceinline__ Fq G1_CURVE_B_MONT() {
  const Fq one = Fq::one(), two = ff_dbl(one);
#ifdef MH_CURVE_BN254
  return ff_add(two, one);
#else
  return ff_dbl(two);
#endif
}

struct G1Xyzz {     // 192 bytes
  Fq x, y, zz, zzz;
  static __device__ __forceinline__ G1Xyzz identity() {
    G1Xyzz r;
    r.x = Fq::zero(); r.y = Fq::zero(); r.zz = Fq::zero(); r.zzz = Fq::zero();
    return r;
  }
  __device__ __forceinline__ bool is_identity() const { return zz.is_zero(); }
};

__device__ __forceinline__ G1Affine g1_load_affine(const G1Affine* p) {
  G1Affine r;
  r.x = ff_load(&p->x);
  r.y = ff_load(&p->y);
  return r;
}
__device__ __forceinline__ G1Xyzz g1_load_xyzz(const G1Xyzz* p) {
  G1Xyzz r;
  r.x = ff_load(&p->x); r.y = ff_load(&p->y); r.zz = ff_load(&p->zz); r.zzz = ff_load(&p->zzz);
  return r;
}
__device__ __forceinline__ void g1_store_xyzz(G1Xyzz* p, const G1Xyzz& a) {
  ff_store(&p->x, a.x); ff_store(&p->y, a.y); ff_store(&p->zz, a.zz); ff_store(&p->zzz, a.zzz);
}

// 2 * (affine point)   [EFD mdbl-2008-s-1, a = 0]
__device__ __noinline__ void g1_dbl_affine(G1Xyzz& r, const Fq& x1, const Fq& y1) {
  Fq U = ff_dbl(y1);
  Fq V = ff_sqr(U);
  Fq W = ff_mul(U, V);
  Fq S = ff_mul(x1, V);
  Fq xx = ff_sqr(x1);
  Fq M = ff_add(ff_dbl(xx), xx);
  Fq X3 = ff_sub(ff_sqr(M), ff_dbl(S));
  Fq Y3 = ff_sub(ff_mul(M, ff_sub(S, X3)), ff_mul(W, y1));
  r.x = X3; r.y = Y3; r.zz = V; r.zzz = W;
}

// acc += (x2, y2) affine, with optional negation of the affine point applied by
// the caller.  [EFD madd-2008-s]  8M + 2S.
__device__ __forceinline__ void g1_madd(G1Xyzz& acc, const Fq& x2, const Fq& y2) {
  if (acc.is_identity()) {
    acc.x = x2; acc.y = y2; acc.zz = Fq::one(); acc.zzz = Fq::one();
    return;
  }
  Fq U2 = ff_mul(x2, acc.zz);
  Fq S2 = ff_mul(y2, acc.zzz);
  Fq Pp = ff_sub(U2, acc.x);
  Fq Rr = ff_sub(S2, acc.y);
  if (Pp.is_zero()) {
    if (Rr.is_zero()) { g1_dbl_affine(acc, x2, y2); }
    else { acc = G1Xyzz::identity(); }
    return;
  }
  Fq PP = ff_sqr(Pp);
  Fq PPP = ff_mul(Pp, PP);
  Fq Q = ff_mul(acc.x, PP);
  Fq X3 = ff_sub(ff_sub(ff_sqr(Rr), PPP), ff_dbl(Q));
  Fq Y3 = ff_sub(ff_mul(Rr, ff_sub(Q, X3)), ff_mul(acc.y, PPP));
  acc.zz = ff_mul(acc.zz, PP);
  acc.zzz = ff_mul(acc.zzz, PPP);
  acc.x = X3; acc.y = Y3;
}

// r = 2 * a   [EFD dbl-2008-s-1, a = 0]
__device__ __noinline__ void g1_dbl(G1Xyzz& a) {
  if (a.is_identity()) return;
  Fq U = ff_dbl(a.y);
  Fq V = ff_sqr(U);
  Fq W = ff_mul(U, V);
  Fq S = ff_mul(a.x, V);
  Fq xx = ff_sqr(a.x);
  Fq M = ff_add(ff_dbl(xx), xx);
  Fq X3 = ff_sub(ff_sqr(M), ff_dbl(S));
  Fq Y3 = ff_sub(ff_mul(M, ff_sub(S, X3)), ff_mul(W, a.y));
  a.zz = ff_mul(V, a.zz);
  a.zzz = ff_mul(W, a.zzz);
  a.x = X3; a.y = Y3;
}

// acc += b   [EFD add-2008-s]  12M + 2S
__device__ __noinline__ void g1_add(G1Xyzz& acc, const G1Xyzz& b) {
  if (b.is_identity()) return;
  if (acc.is_identity()) { acc = b; return; }
  Fq U1 = ff_mul(acc.x, b.zz);
  Fq U2 = ff_mul(b.x, acc.zz);
  Fq S1 = ff_mul(acc.y, b.zzz);
  Fq S2 = ff_mul(b.y, acc.zzz);
  Fq Pp = ff_sub(U2, U1);
  Fq Rr = ff_sub(S2, S1);
  if (Pp.is_zero()) {
    if (Rr.is_zero()) { g1_dbl(acc); }
    else { acc = G1Xyzz::identity(); }
    return;
  }
  Fq PP = ff_sqr(Pp);
  Fq PPP = ff_mul(Pp, PP);
  Fq Q = ff_mul(U1, PP);
  Fq X3 = ff_sub(ff_sub(ff_sqr(Rr), PPP), ff_dbl(Q));
  Fq Y3 = ff_sub(ff_mul(Rr, ff_sub(Q, X3)), ff_mul(S1, PPP));
  acc.zz = ff_mul(ff_mul(acc.zz, b.zz), PP);
  acc.zzz = ff_mul(ff_mul(acc.zzz, b.zzz), PPP);
  acc.x = X3; acc.y = Y3;
}
