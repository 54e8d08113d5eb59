// Host-side (x86-64) Montgomery arithmetic for the curve's Fr / Fq (BLS12-381, or BN254) and G1 group
// operations, used by the product's host code for the O(1)-per-call work around
// the kernels: combining MSM window sums, affine normalisation, Fiat-Shamir
// challenge arithmetic, n^-1 for inverse NTTs.  64-bit limbs, same in-memory
// layout as arkworks' BigInteger256/384 (little-endian u64 limbs, Montgomery
// form) so buffers cross the C ABI without conversion.
//
// This is product code (not the oracle): it never runs per-element work that
// the reference's hot path would run on its cores; the parity oracle lives in
// /oracle and is independent of this file.
#pragma once
#include <stdint.h>
#include <string.h>

namespace hostff {

typedef unsigned __int128 u128;

template <class P>
struct HFp {
  static constexpr int N = P::N;
  uint64_t v[P::N];

  static HFp zero() { HFp r; memset(r.v, 0, sizeof(r.v)); return r; }
  static uint64_t MOD_LIMB(int i) { return P::MOD[i]; }
  static HFp one() { HFp r; memcpy(r.v, P::ONE, sizeof(r.v)); return r; }
  bool is_zero() const { uint64_t o = 0; for (int i = 0; i < N; i++) o |= v[i]; return o == 0; }
  bool operator==(const HFp& b) const { return memcmp(v, b.v, sizeof(v)) == 0; }
  bool operator!=(const HFp& b) const { return !(*this == b); }

  static bool geq_mod(const uint64_t* a) {
    for (int i = N - 1; i >= 0; i--) {
      if (a[i] > P::MOD[i]) return true;
      if (a[i] < P::MOD[i]) return false;
    }
    return true;
  }
  static void sub_mod(uint64_t* a) {
    uint64_t borrow = 0;
    for (int i = 0; i < N; i++) {
      u128 d = (u128)a[i] - P::MOD[i] - borrow;
      a[i] = (uint64_t)d;
      borrow = (uint64_t)(d >> 127);
    }
  }
  HFp operator+(const HFp& b) const {
    HFp r; uint64_t c = 0;
    for (int i = 0; i < N; i++) { u128 s = (u128)v[i] + b.v[i] + c; r.v[i] = (uint64_t)s; c = (uint64_t)(s >> 64); }
    if (c || geq_mod(r.v)) sub_mod(r.v);
    return r;
  }
  HFp operator-(const HFp& b) const {
    HFp r; uint64_t borrow = 0;
    for (int i = 0; i < N; i++) { u128 d = (u128)v[i] - b.v[i] - borrow; r.v[i] = (uint64_t)d; borrow = (uint64_t)(d >> 127); }
    if (borrow) { uint64_t c = 0; for (int i = 0; i < N; i++) { u128 s = (u128)r.v[i] + P::MOD[i] + c; r.v[i] = (uint64_t)s; c = (uint64_t)(s >> 64); } }
    return r;
  }
  HFp neg() const { return is_zero() ? *this : zero() - *this; }
  HFp dbl() const { return *this + *this; }
  HFp operator*(const HFp& b) const {
    uint64_t t[N + 2];
    memset(t, 0, sizeof(t));
    for (int i = 0; i < N; i++) {
      uint64_t c = 0;
      for (int j = 0; j < N; j++) { u128 p = (u128)v[i] * b.v[j] + t[j] + c; t[j] = (uint64_t)p; c = (uint64_t)(p >> 64); }
      u128 s = (u128)t[N] + c; t[N] = (uint64_t)s; t[N + 1] = (uint64_t)(s >> 64);
      uint64_t m = t[0] * P::INV;
      u128 p = (u128)m * P::MOD[0] + t[0]; c = (uint64_t)(p >> 64);
      for (int j = 1; j < N; j++) { p = (u128)m * P::MOD[j] + t[j] + c; t[j - 1] = (uint64_t)p; c = (uint64_t)(p >> 64); }
      s = (u128)t[N] + c; t[N - 1] = (uint64_t)s; t[N] = t[N + 1] + (uint64_t)(s >> 64);
    }
    HFp r; memcpy(r.v, t, sizeof(r.v));
    if (t[N] || geq_mod(r.v)) sub_mod(r.v);
    return r;
  }
  HFp sqr() const { return (*this) * (*this); }
  HFp pow(const uint64_t* e, int nlimbs) const {
    HFp acc = one();
    for (int i = nlimbs - 1; i >= 0; i--)
      for (int b = 63; b >= 0; b--) { acc = acc.sqr(); if ((e[i] >> b) & 1) acc = acc * (*this); }
    return acc;
  }
  HFp pow_u64(uint64_t e) const { return pow(&e, 1); }
  HFp inv_fermat() const {  // 0 -> 0
    uint64_t e[N]; memcpy(e, P::MOD, sizeof(e)); e[0] -= 2;
    return pow(e, N);
  }
  // Inverse by the binary extended Euclidean algorithm on the stored integer (what ark-ff's `inverse` does; 0 -> 0): the Fermat
  // power is ~580 multiplications -- 40 us for Fq on the prover's critical path at every commit round (one per batch_to_affine) and
  // 13 us for each of Fr's ~10 per proof; this is ~1.4 x 381 shift / subtract steps.  The stored integer is a R, so the Euclidean
  // inverse is a^-1 R^-1: two Montgomery multiplications by R^2 bring it to a^-1 R.
  HFp inv() const {
    if (is_zero()) return *this;
    uint64_t u[N], w[N], b[N], c[N];
    memcpy(u, v, sizeof(u)); memcpy(w, P::MOD, sizeof(w));
    memset(b, 0, sizeof(b)); memset(c, 0, sizeof(c)); b[0] = 1;
    auto is_one = [](const uint64_t* a) { uint64_t o = a[0] ^ 1; for (int i = 1; i < N; i++) o |= a[i]; return o == 0; };
    // a >>= k and x = x / 2^k mod p in one pass each (0 < k < 64): x + m p is divisible by 2^k for m = -x p^-1 mod 2^k (P::INV = -p^-1 mod 2^64)
    auto shift_pair = [](uint64_t* a, uint64_t* x, unsigned k) {
      for (int i = 0; i < N - 1; i++) a[i] = (a[i] >> k) | (a[i + 1] << (64 - k));
      a[N - 1] >>= k;
      const uint64_t m = (x[0] * P::INV) & ((1ull << k) - 1);
      u128 t = (u128)m * P::MOD[0] + x[0];
      uint64_t lo = (uint64_t)t, cy = (uint64_t)(t >> 64);
      for (int i = 1; i < N; i++) {
        t = (u128)m * P::MOD[i] + x[i] + cy;
        const uint64_t nx = (uint64_t)t; cy = (uint64_t)(t >> 64);
        x[i - 1] = (lo >> k) | (nx << (64 - k));
        lo = nx;
      }
      x[N - 1] = (lo >> k) | (cy << (64 - k));
    };
    auto geq = [](const uint64_t* a, const uint64_t* d) { for (int i = N - 1; i >= 0; i--) { if (a[i] != d[i]) return a[i] > d[i]; } return true; };
    auto sub = [](uint64_t* a, const uint64_t* d) { uint64_t bw = 0; for (int i = 0; i < N; i++) { u128 t = (u128)a[i] - d[i] - bw; a[i] = (uint64_t)t; bw = (uint64_t)(t >> 127); } return bw; };
    auto sub_modp = [&](uint64_t* a, const uint64_t* d) {      // a = a - d mod p, both < p
      if (sub(a, d)) { uint64_t cy = 0; for (int i = 0; i < N; i++) { u128 t = (u128)a[i] + P::MOD[i] + cy; a[i] = (uint64_t)t; cy = (uint64_t)(t >> 64); } }
    };
    while (!is_one(u) && !is_one(w)) {
      while (!(u[0] & 1)) { const unsigned k = u[0] ? (unsigned)__builtin_ctzll(u[0]) : 63u; shift_pair(u, b, k); }
      while (!(w[0] & 1)) { const unsigned k = w[0] ? (unsigned)__builtin_ctzll(w[0]) : 63u; shift_pair(w, c, k); }
      if (geq(u, w)) { sub(u, w); sub_modp(b, c); } else { sub(w, u); sub_modp(c, b); }
    }
    HFp x, r2;
    memcpy(x.v, is_one(u) ? b : c, sizeof(x.v)); memcpy(r2.v, P::R2, sizeof(r2.v));
    return (x * r2) * r2;
  }
  // canonical integer (little-endian limbs) <-> Montgomery
  static HFp from_canonical(const uint64_t* c) { HFp a; memcpy(a.v, c, sizeof(a.v)); HFp r2; memcpy(r2.v, P::R2, sizeof(r2.v)); return a * r2; }
  void to_canonical(uint64_t* out) const { HFp o = zero(); o.v[0] = 1; HFp r = (*this) * o; memcpy(out, r.v, sizeof(r.v)); }
  static HFp from_u64(uint64_t x) { uint64_t c[N]; memset(c, 0, sizeof(c)); c[0] = x; return from_canonical(c); }
};

struct FrP {
  static constexpr int N = 4;
  static constexpr uint64_t MOD[4] = {0xffffffff00000001ull, 0x53bda402fffe5bfeull, 0x3339d80809a1d805ull, 0x73eda753299d7d48ull};
  static constexpr uint64_t INV = 0xfffffffeffffffffull;
  static constexpr uint64_t ONE[4] = {0x00000001fffffffeull, 0x5884b7fa00034802ull, 0x998c4fefecbc4ff5ull, 0x1824b159acc5056full};
  static constexpr uint64_t R2[4] = {0xc999e990f3f29c6dull, 0x2b6cedcb87925c23ull, 0x05d314967254398full, 0x0748d9d99f59ff11ull};
};
struct FqP {
  static constexpr int N = 6;
  static constexpr uint64_t MOD[6] = {0xb9feffffffffaaabull, 0x1eabfffeb153ffffull, 0x6730d2a0f6b0f624ull,
                                      0x64774b84f38512bfull, 0x4b1ba7b6434bacd7ull, 0x1a0111ea397fe69aull};
  static constexpr uint64_t INV = 0x89f3fffcfffcfffdull;
  static constexpr uint64_t ONE[6] = {0x760900000002fffdull, 0xebf4000bc40c0002ull, 0x5f48985753c758baull,
                                      0x77ce585370525745ull, 0x5c071a97a256ec6dull, 0x15f65ec3fa80e493ull};
  static constexpr uint64_t R2[6] = {0xf4df1f341c341746ull, 0x0a76e6a609d104f1ull, 0x8de5476c4c95b6d5ull,
                                     0x67eb88a9939d83c0ull, 0x9a793e85b519952dull, 0x11988fe592cae3aaull};
};
struct Bn254FrP {
  static constexpr int N = 4;
  static constexpr uint64_t MOD[4] = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
  static constexpr uint64_t INV = 0xc2e1f593efffffffull;
  static constexpr uint64_t ONE[4] = {0xac96341c4ffffffbull, 0x36fc76959f60cd29ull, 0x666ea36f7879462eull, 0x0e0a77c19a07df2full};
  static constexpr uint64_t R2[4] = {0x1bb8e645ae216da7ull, 0x53fe3ab1e35c59e3ull, 0x8c49833d53bb8085ull, 0x0216d0b17f4e44a5ull};
};
struct Bn254FqP {
  static constexpr int N = 4;
  static constexpr uint64_t MOD[4] = {0x3c208c16d87cfd47ull, 0x97816a916871ca8dull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
  static constexpr uint64_t INV = 0x87d20782e4866389ull;
  static constexpr uint64_t ONE[4] = {0xd35d438dc58f0d9dull, 0x0a78eb28f5c70b3dull, 0x666ea36f7879462cull, 0x0e0a77c19a07df2full};
  static constexpr uint64_t R2[4] = {0xf32cfc5b538afa89ull, 0xb5e71911d44501fbull, 0x47ab1eff0a417ff6ull, 0x06d89f71cab8351full};
};

// build-time curve choice (see ff.cuh)
#ifdef MH_CURVE_BN254
typedef HFp<Bn254FrP> HFr;
typedef HFp<Bn254FqP> HFq;
constexpr int CURVE_ID = 1;
constexpr uint32_t FR_TWO_ADICITY_H = 28;
constexpr uint64_t FR_SHAVE_MASK_TOP64 = 0x3fffffffffffffffull;
constexpr uint64_t G1_B = 3;                 // y^2 = x^3 + 3
constexpr uint64_t FR_MULT_GENERATOR = 5;    // F::multiplicative_generator()
// 2^28-th root of unity of Fr (5^((r-1)/2^28)), Montgomery form
static inline HFr fr_two_adic_root() {
  static const uint64_t c[4] = {0x9bd61b6e725b19f0ull, 0x402d111e41112ed4ull, 0x00e0a7eb8ef62abcull, 0x2a3c09f0a58a7e85ull};
  return HFr::from_canonical(c);
}
static inline void g1_generator_canonical(uint64_t* gx, uint64_t* gy) { memset(gx, 0, 32); memset(gy, 0, 32); gx[0] = 1; gy[0] = 2; }
#else
typedef HFp<FrP> HFr;
typedef HFp<FqP> HFq;
constexpr int CURVE_ID = 0;
constexpr uint32_t FR_TWO_ADICITY_H = 32;
constexpr uint64_t FR_SHAVE_MASK_TOP64 = 0x7fffffffffffffffull;
constexpr uint64_t G1_B = 4;                 // y^2 = x^3 + 4
constexpr uint64_t FR_MULT_GENERATOR = 7;    // F::multiplicative_generator()
// 2^32-th root of unity of Fr (7^((r-1)/2^32)), Montgomery form
static inline HFr fr_two_adic_root() {
  static const uint64_t c[4] = {0x3829971f439f0d2bull, 0xb63683508c2280b9ull, 0xd09b681922c813b4ull, 0x16a2a19edfe81f20ull};
  return HFr::from_canonical(c);
}
static inline void g1_generator_canonical(uint64_t* gx, uint64_t* gy) {
  static const uint64_t x[6] = {0xfb3af00adb22c6bbull, 0x6c55e83ff97a1aefull, 0xa14e3a3f171bac58ull,
                                0xc3688c4f9774b905ull, 0x2695638c4fa9ac0full, 0x17f1d3a73197d794ull};
  static const uint64_t y[6] = {0x0caa232946c5e7e1ull, 0xd03cc744a2888ae4ull, 0x00db18cb2c04b3edull,
                                0xfcf5e095d5d00af6ull, 0xa09e30ed741d8ae4ull, 0x08b3f481e3aaa0f1ull};
  memcpy(gx, x, 48); memcpy(gy, y, 48);
}
#endif
// sizes that depend on the curve (64-bit limbs / bytes)
constexpr int FQ_L = HFq::N;                 // limbs per Fq
constexpr size_t FQ_B = 8 * HFq::N;          // bytes per Fq
constexpr size_t PT_B = 2 * FQ_B;            // bytes per affine point
constexpr int AFF_L = 2 * FQ_L, XYZ_L = 3 * FQ_L, XYZZ_L = 4 * FQ_L;

// ---- G1 (Jacobian X,Y,Z; x = X/Z^2, y = Y/Z^3; Z = 0 identity) -- the layout of
// arkworks' G1Projective.
struct HG1Affine { HFq x, y; bool inf; };
struct HG1 {
  HFq X, Y, Z;
  static HG1 identity() { HG1 r; r.X = HFq::one(); r.Y = HFq::one(); r.Z = HFq::zero(); return r; }
  bool is_identity() const { return Z.is_zero(); }
  static HG1 from_affine(const HG1Affine& a) { if (a.inf) return identity(); HG1 r; r.X = a.x; r.Y = a.y; r.Z = HFq::one(); return r; }
  // from device XYZZ (x = X/ZZ, y = Y/ZZZ): pick Z = ZZ*ZZZ... no inversion needed:
  // Z' = ZZZ*ZZ has Z'^2 = ZZ^2 ZZZ^2, Z'^3 = ZZ^3 ZZZ^3; X' = X*ZZ*ZZZ^2, Y' = Y*ZZ^3*ZZZ^2.
  static HG1 from_xyzz(const HFq& X, const HFq& Y, const HFq& ZZ, const HFq& ZZZ) {
    if (ZZ.is_zero()) return identity();
    HG1 r;
    HFq zzz2 = ZZZ.sqr();
    r.X = X * ZZ * zzz2;
    r.Y = Y * ZZ.sqr() * ZZ * zzz2;
    r.Z = ZZ * ZZZ;
    return r;
  }
  HG1 dbl() const {
    if (is_identity()) return *this;
    HFq A = X.sqr(), B = Y.sqr(), C = B.sqr();
    HFq D = ((X + B).sqr() - A - C).dbl();
    HFq E = A.dbl() + A;
    HFq F = E.sqr();
    HG1 r;
    r.X = F - D.dbl();
    r.Y = E * (D - r.X) - C.dbl().dbl().dbl();
    r.Z = (Y * Z).dbl();
    return r;
  }
  HG1 add(const HG1& b) const {
    if (is_identity()) return b;
    if (b.is_identity()) return *this;
    HFq Z1Z1 = Z.sqr(), Z2Z2 = b.Z.sqr();
    HFq U1 = X * Z2Z2, U2 = b.X * Z1Z1;
    HFq S1 = Y * b.Z * Z2Z2, S2 = b.Y * Z * Z1Z1;
    if (U1 == U2) { if (S1 == S2) return dbl(); return identity(); }
    HFq H = U2 - U1, Rr = S2 - S1;
    HFq HH = H.sqr(), HHH = H * HH, V = U1 * HH;
    HG1 r;
    r.X = Rr.sqr() - HHH - V.dbl();
    r.Y = Rr * (V - r.X) - S1 * HHH;
    r.Z = Z * b.Z * H;
    return r;
  }
  HG1 neg() const { HG1 r = *this; r.Y = Y.neg(); return r; }
  HG1 mul(const uint64_t* k, int nlimbs) const {  // canonical scalar
    HG1 acc = identity();
    for (int i = nlimbs - 1; i >= 0; i--)
      for (int b = 63; b >= 0; b--) { acc = acc.dbl(); if ((k[i] >> b) & 1) acc = acc.add(*this); }
    return acc;
  }
  HG1Affine to_affine() const {
    HG1Affine a;
    if (is_identity()) { a.x = HFq::zero(); a.y = HFq::zero(); a.inf = true; return a; }
    HFq zi = Z.inv(), zi2 = zi.sqr();
    a.x = X * zi2; a.y = Y * zi2 * zi; a.inf = false;
    return a;
  }
};

// Affine normalisation of several points with ONE field inversion (Montgomery's trick): a commit round yields 3-5
// commitments and the host's Fermat inversion is ~50 us, during which the GPU waits for the next Fiat-Shamir challenge.
inline void batch_to_affine(const HG1* pts, size_t n, HG1Affine* out) {
  HFq prefix[16];
  HFq acc = HFq::one();
  if (n > 16) { for (size_t i = 0; i < n; i++) out[i] = pts[i].to_affine(); return; }
  for (size_t i = 0; i < n; i++) {
    prefix[i] = acc;
    if (!pts[i].is_identity()) acc = acc * pts[i].Z;
  }
  HFq inv = acc.inv();
  for (size_t i = n; i-- > 0;) {
    if (pts[i].is_identity()) { out[i].x = HFq::zero(); out[i].y = HFq::zero(); out[i].inf = true; continue; }
    const HFq zi = inv * prefix[i], zi2 = zi.sqr();
    inv = inv * pts[i].Z;
    out[i].x = pts[i].X * zi2; out[i].y = pts[i].Y * zi2 * zi; out[i].inf = false;
  }
}

}  // namespace hostff
