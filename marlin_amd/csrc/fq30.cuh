// Base-field arithmetic on 30-bit limbs for the fixed-base bucket accumulation (msm_fb.cuh).
//
// Why a second representation: on gfx950 every VALU instruction issues at about the same rate
// (profiles/r01_microbench.txt: v_mad_u64_u32 30 T/s, v_addc 33 T/s), so a field multiplication costs what its
// instruction count says.  With 32-bit limbs (ff.cuh) each of the 288 limb products needs a v_mad_u64_u32 AND a carry
// instruction, because a 64-bit accumulator overflows after two products.  With 30-bit limbs 13 products fit in a
// 64-bit accumulator: the columns of a 13 x 13 product are plain sums of v_mad_u64_u32 results, carries are taken
// once per column (and, shift), and the whole multiplication is ordinary C++ that compiles to ~530 instructions
// instead of ~700.  Values are Montgomery residues for R' = 2^(30 NL) (2^390 for BLS12-381) and are kept LAZILY
// reduced: R' / p ~ 630, so a product of two values below ~20 p is again below 2 p and additions / subtractions
// need no conditional correction at all (subtractions add a fixed multiple of p instead).
//
// Used only inside the accumulate kernel; buckets are converted back to the 32-bit Montgomery form of ff.cuh when
// they are stored (one extra multiplication per coordinate and bucket).
#pragma once
#include "ff.cuh"
#include "fq30_consts.inc"

#ifdef MH_CURVE_BN254
using Fq30Params = Fq30Params_BN254;
#else
using Fq30Params = Fq30Params_BLS12_381;
#endif

struct Fq30 {
  static constexpr int NL = Fq30Params::NL;
  u32 v[NL];      // limbs < 2^30 (the top limb holds whatever is left: values stay far below 2^(30 NL))
};

constexpr u32 M30 = (1u << 30) - 1;
#include "fq30_mul_gen.inc"

// Montgomery reduction of a double-length value given as 2 NL normalised 30-bit limbs: t / R' mod p, lazily reduced.
__device__ __forceinline__ Fq30 f30_redc_cxx(const u32* t) {
  constexpr int NL = Fq30::NL;
  using PP = Fq30Params;
  u32 m[NL];
  Fq30 r;
  u64 acc = 0;
  // column by column: m_k clears the low 30 bits of column k
#pragma unroll
  for (int k = 0; k < NL; k++) {
    acc += t[k];
#pragma unroll
    for (int i = 0; i < k; i++) acc += (u64)m[i] * PP::P[k - i];
    m[k] = ((u32)acc * PP::PINV) & M30;
    acc += (u64)m[k] * PP::P[0];
    acc >>= 30;
  }
#pragma unroll
  for (int k = NL; k < 2 * NL; k++) {
    acc += t[k];
#pragma unroll
    for (int i = k - NL + 1; i < NL; i++) acc += (u64)m[i] * PP::P[k - i];
    if (k < 2 * NL - 1) { r.v[k - NL] = (u32)acc & M30; acc >>= 30; }
    else r.v[k - NL] = (u32)acc;
  }
  return r;
}

// a * b / R' mod p, lazily reduced: for a, b < 20 p the result is < 2 p.  Limbs of a and b must be < 2^30.
__device__ __forceinline__ Fq30 f30_mul_cxx(const Fq30& a, const Fq30& b) {
  constexpr int NL = Fq30::NL;
  u32 t[2 * NL];
  u64 acc = 0;
  // product columns: at most NL products of < 2^60 each plus a carry < 2^35
#pragma unroll
  for (int k = 0; k < 2 * NL - 1; k++) {
#pragma unroll
    for (int i = (k < NL ? 0 : k - NL + 1); i <= (k < NL ? k : NL - 1); i++) acc += (u64)a.v[i] * b.v[k - i];
    t[k] = (u32)acc & M30;
    acc >>= 30;
  }
  t[2 * NL - 1] = (u32)acc;
  return f30_redc_cxx(t);
}
// a^2 / R': the cross products once, against the doubled limbs (2 a_i < 2^31: a column still sums to < 2^64)
__device__ __forceinline__ Fq30 f30_sqr_cxx(const Fq30& a) {
  constexpr int NL = Fq30::NL;
  u32 a2[NL];
#pragma unroll
  for (int i = 0; i < NL; i++) a2[i] = a.v[i] << 1;
  u32 t[2 * NL];
  u64 acc = 0;
#pragma unroll
  for (int k = 0; k < 2 * NL - 1; k++) {
#pragma unroll
    for (int i = (k < NL ? 0 : k - NL + 1); 2 * i < k; i++) acc += (u64)a2[i] * a.v[k - i];
    if ((k & 1) == 0) acc += (u64)a.v[k >> 1] * a.v[k >> 1];
    t[k] = (u32)acc & M30;
    acc >>= 30;
  }
  t[2 * NL - 1] = (u32)acc;
  return f30_redc_cxx(t);
}

// The versions the kernels use: the same column sums and reduction, generated with ONE inline-asm statement of
// v_mad_u64_u32 per column (fq30_mul_gen.inc): hipcc otherwise splits every column into several accumulator chains and
// joins them with ~40 extra 64-bit additions per multiplication.  Limb-for-limb identical to the *_cxx versions
// (checked by mh_selftest_fq30).
#ifdef MH_CURVE_BN254
#define F30_GEN(fn) fn##_BN254
#else
#define F30_GEN(fn) fn##_BLS12_381
#endif
// separate product and reduction passes (the round-1 form; kept as the cross-check of the fused form below)
__device__ __forceinline__ Fq30 f30_mul_sep(const Fq30& a, const Fq30& b) {
  u32 t[2 * Fq30::NL];
  Fq30 r;
  F30_GEN(f30_cols_mul)(t, a.v, b.v);
  F30_GEN(f30_redc)(r.v, t);
  return r;
}
__device__ __forceinline__ Fq30 f30_sqr_sep(const Fq30& a) {
  u32 t[2 * Fq30::NL];
  Fq30 r;
  F30_GEN(f30_cols_sqr)(t, a.v);
  F30_GEN(f30_redc)(r.v, t);
  return r;
}
// What the kernels use: product and reduction columns summed in ONE accumulator (finely integrated product scanning,
// gen_fq30.py `fused`): a column costs one mask and one 64-bit shift instead of two masks, two 64-bit shifts and a 64-bit
// add; 5 of the 26 columns of BLS12-381 move an early carry out to stay below 2^64.  Limb-for-limb equal to the separate
// form and to the *_cxx versions (mh_selftest_fq30).
__device__ __forceinline__ Fq30 f30_mul(const Fq30& a, const Fq30& b) {
  Fq30 r;
  F30_GEN(f30_mulredc)(r.v, a.v, b.v);
  return r;
}
__device__ __forceinline__ Fq30 f30_sqr(const Fq30& a) {
  Fq30 r;
  F30_GEN(f30_sqrredc)(r.v, a.v);
  return r;
}
// (a b + c d) / R' with ONE Montgomery reduction: both products' columns go into the same accumulator.  For
// a b + c d < 64 p^2 the result is below 1.2 p.
__device__ __forceinline__ Fq30 f30_mul2(const Fq30& a, const Fq30& b, const Fq30& c, const Fq30& d) {
  Fq30 r;
  F30_GEN(f30_mul2redc)(r.v, a.v, b.v, c.v, d.v);
  return r;
}

// Two or three INDEPENDENT multiplications as interleaved dependency chains (gen_fq30.py `fused_multi`): for kernels that run one
// wave per SIMD (the bucket reduction), where a single multiplication's chain of ~400 dependent instructions leaves the VALU idle
// between issues.  Limb for limb the results of f30_mul / f30_sqr / f30_mul2 (mh_selftest_fq30).  Outputs may alias inputs.
__device__ __forceinline__ void f30_mul_x2(Fq30& r0, const Fq30& a0, const Fq30& b0, Fq30& r1, const Fq30& a1, const Fq30& b1) {
  Fq30 t0, t1;
  F30_GEN(f30_multi_mul_mul)(t0.v, a0.v, b0.v, t1.v, a1.v, b1.v);
  r0 = t0; r1 = t1;
}
__device__ __forceinline__ void f30_mul_x3(Fq30& r0, const Fq30& a0, const Fq30& b0, Fq30& r1, const Fq30& a1, const Fq30& b1,
                                           Fq30& r2, const Fq30& a2, const Fq30& b2) {
  Fq30 t0, t1, t2;
  F30_GEN(f30_multi_mul_mul_mul)(t0.v, a0.v, b0.v, t1.v, a1.v, b1.v, t2.v, a2.v, b2.v);
  r0 = t0; r1 = t1; r2 = t2;
}
__device__ __forceinline__ void f30_sqr_x2(Fq30& r0, const Fq30& a0, Fq30& r1, const Fq30& a1) {
  Fq30 t0, t1;
  F30_GEN(f30_multi_sqr_sqr)(t0.v, a0.v, t1.v, a1.v);
  r0 = t0; r1 = t1;
}
// r0 = a0^2, r1 = a1 b1
__device__ __forceinline__ void f30_sqr_mul(Fq30& r0, const Fq30& a0, Fq30& r1, const Fq30& a1, const Fq30& b1) {
  Fq30 t0, t1;
  F30_GEN(f30_multi_sqr_mul)(t0.v, a0.v, t1.v, a1.v, b1.v);
  r0 = t0; r1 = t1;
}
// r0 = (a0 b0 + c0 d0) under one reduction, r1 = a1 b1
__device__ __forceinline__ void f30_mul2_mul(Fq30& r0, const Fq30& a0, const Fq30& b0, const Fq30& c0, const Fq30& d0,
                                             Fq30& r1, const Fq30& a1, const Fq30& b1) {
  Fq30 t0, t1;
  F30_GEN(f30_multi_mul2_mul)(t0.v, a0.v, b0.v, c0.v, d0.v, t1.v, a1.v, b1.v);
  r0 = t0; r1 = t1;
}

// a + b (no reduction)
__device__ __forceinline__ Fq30 f30_add(const Fq30& a, const Fq30& b) {
  Fq30 r;
  u32 c = 0;
#pragma unroll
  for (int i = 0; i < Fq30::NL; i++) {
    u32 s = a.v[i] + b.v[i] + c;
    if (i < Fq30::NL - 1) { r.v[i] = s & M30; c = s >> 30; } else r.v[i] = s;
  }
  return r;
}
// a - b + K p for the generated multiples K in {2, 3, 4, 8}; requires b <= K p
template <int K>
__device__ __forceinline__ Fq30 f30_sub(const Fq30& a, const Fq30& b) {
  using PP = Fq30Params;
  Fq30 r;
  int c = 0;
#pragma unroll
  for (int i = 0; i < Fq30::NL; i++) {
    const u32 kp = K == 2 ? PP::P2[i] : (K == 3 ? PP::P3[i] : (K == 4 ? PP::P4[i] : PP::P8[i]));
    int s = (int)(a.v[i] - b.v[i]) + (int)kp + c;        // in (-2^30, 2^31)
    if (i < Fq30::NL - 1) { r.v[i] = (u32)s & M30; c = s >> 30; } else r.v[i] = (u32)s;
  }
  return r;
}
__device__ __forceinline__ Fq30 f30_dbl(const Fq30& a) { return f30_add(a, a); }
// a - 2 b + K p in one carry pass; requires 2 b <= K p
template <int K>
__device__ __forceinline__ Fq30 f30_sub2(const Fq30& a, const Fq30& b) {
  using PP = Fq30Params;
  Fq30 r;
  int c = 0;
#pragma unroll
  for (int i = 0; i < Fq30::NL; i++) {
    const u32 kp = K == 2 ? PP::P2[i] : (K == 3 ? PP::P3[i] : (K == 4 ? PP::P4[i] : PP::P8[i]));
    int s = (int)(a.v[i] - 2 * b.v[i]) + (int)kp + c;      // in [-2^31, 2^31)
    if (i < Fq30::NL - 1) { r.v[i] = (u32)s & M30; c = s >> 30; } else r.v[i] = (u32)s;
  }
  return r;
}

// a == 0 mod p for a normalised value below 16 p: a = j p implies a * p^-1 = j mod 2^30, so one multiplication rejects
// all but ~2^-26 of the non-zero values, and those are compared with j p limb by limb (exact: a collision-free proof
// leaves the accumulate kernel's deferred list empty and the fix-up pass returns at once).
__device__ __forceinline__ bool f30_is_zero(const Fq30& a) {
  const u32 j = (a.v[0] * Fq30Params::PINV_POS) & M30;
  if (__builtin_expect(j >= 16u, 1)) return false;
  u64 c = 0;
  bool eq = true;
#pragma unroll
  for (int i = 0; i < Fq30::NL; i++) {
    c += (u64)j * Fq30Params::P[i];
    if (i < Fq30::NL - 1) { eq = eq && a.v[i] == ((u32)c & M30); c >>= 30; }
    else eq = eq && a.v[i] == (u32)c;
  }
  return eq;
}

// ---- conversions to / from the 32-bit Montgomery form of ff.cuh --------------------------------------------
__device__ __forceinline__ Fq30 f30_split(const Fq& x) {       // plain regrouping of the bits
  Fq30 r;
#pragma unroll
  for (int i = 0; i < Fq30::NL; i++) {
    const int bit = 30 * i, w = bit >> 5, sh = bit & 31;
    u64 two = w < Fq::N ? x.v[w] : 0;
    if (w + 1 < Fq::N) two |= (u64)x.v[w + 1] << 32;
    r.v[i] = (u32)(two >> sh) & M30;
  }
  return r;
}
__device__ __forceinline__ Fq f30_pack(const Fq30& a) {        // limbs must be normalised and the value < 2^(32 N)
  Fq r;
#pragma unroll
  for (int w = 0; w < Fq::N; w++) {
    const int bit = 32 * w, i = bit / 30, sh = bit - 30 * i;   // limb i contributes from its bit sh
    u64 v = (u64)a.v[i] >> sh;
    if (i + 1 < Fq30::NL) v |= (u64)a.v[i + 1] << (30 - sh);
    if (i + 2 < Fq30::NL) v |= (u64)a.v[i + 2] << (60 - sh);
    r.v[w] = (u32)v;
  }
  return r;
}
// standard Montgomery residue -> 30-bit Montgomery residue (exactly reduced)
__device__ __forceinline__ Fq30 f30_from_fq(const Fq& x) {
  Fq k;
#pragma unroll
  for (int i = 0; i < Fq::N; i++) k.v[i] = Fq30Params::TO30[i];
  return f30_split(ff_mul(x, k));
}
// lazily reduced 30-bit residue (< 20 p, limbs normalised) -> standard Montgomery residue < p.  A multiplication by
// one (in the R' form) first brings the value below 2 p, which fits the 32-bit limbs of either curve (9 p does not
// fit 256 bits for BN254).
__device__ __forceinline__ Fq f30_to_fq(const Fq30& a) {
  Fq30 one;
#pragma unroll
  for (int i = 0; i < Fq30::NL; i++) one.v[i] = Fq30Params::ONE[i];
  Fq k;
#pragma unroll
  for (int i = 0; i < Fq::N; i++) k.v[i] = Fq30Params::FROM30[i];
  return ff_mul(f30_pack(f30_mul(a, one)), k);
}
