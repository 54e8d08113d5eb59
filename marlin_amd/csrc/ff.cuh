// Device-side prime-field arithmetic for gfx950 (CDNA4): BLS12-381 Fr (8 x u32)
// and Fq (12 x u32), Montgomery form, little-endian 32-bit limbs (bit-identical
// in memory to arkworks' 4 x / 6 x u64 little-endian BigInteger layout on a
// little-endian host).
//
// Replaces, on device, what the reference obtains from ark-ff 0.3 `Fp256` /
// `Fp384` (third-party; call sites SURVEY.md §2.2 E4).  Every value handled by
// the kernels stays in Montgomery form, as arkworks keeps it in memory.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint32_t u32;
typedef uint64_t u64;

// ---------------------------------------------------------------------------------
// Field parameters (re-derived in oracle/fields.py and
// cross-checked by tests/test_oracle_fields.py against this header).
// ---------------------------------------------------------------------------------
struct FrParams {
  static constexpr int N = 8;
  static constexpr u32 MOD[8] = {0x00000001u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u,
                                 0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u};
  static constexpr u32 INV = 0xffffffffu;  // -r^-1 mod 2^32
  // R = 2^256 mod r (Montgomery one)
  static constexpr u32 ONE[8] = {0xfffffffeu, 0x00000001u, 0x00034802u, 0x5884b7fau,
                                 0xecbc4ff5u, 0x998c4fefu, 0xacc5056fu, 0x1824b159u};
  // R^2 mod r
  static constexpr u32 R2[8] = {0xf3f29c6du, 0xc999e990u, 0x87925c23u, 0x2b6cedcbu,
                                0x7254398fu, 0x05d31496u, 0x9f59ff11u, 0x0748d9d9u};
};

struct FqParams {
  static constexpr int N = 12;
  static constexpr u32 MOD[12] = {0xffffaaabu, 0xb9feffffu, 0xb153ffffu, 0x1eabfffeu,
                                  0xf6b0f624u, 0x6730d2a0u, 0xf38512bfu, 0x64774b84u,
                                  0x434bacd7u, 0x4b1ba7b6u, 0x397fe69au, 0x1a0111eau};
  static constexpr u32 INV = 0xfffcfffdu;  // -q^-1 mod 2^32
  static constexpr u32 ONE[12] = {0x0002fffdu, 0x76090000u, 0xc40c0002u, 0xebf4000bu,
                                  0x53c758bau, 0x5f489857u, 0x70525745u, 0x77ce5853u,
                                  0xa256ec6du, 0x5c071a97u, 0xfa80e493u, 0x15f65ec3u};
  static constexpr u32 R2[12] = {0x1c341746u, 0xf4df1f34u, 0x09d104f1u, 0x0a76e6a6u,
                                 0x4c95b6d5u, 0x8de5476cu, 0x939d83c0u, 0x67eb88a9u,
                                 0xb519952du, 0x9a793e85u, 0x92cae3aau, 0x11988fe5u};
};

// BN254 (second curve, BASELINE.json configs[4]; ark-bn254 public parameters): r and q are both 254-bit, 8 x u32.
struct Bn254FrParams {
  static constexpr int N = 8;
  static constexpr u32 MOD[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
  static constexpr u32 INV = 0xefffffffu;
  static constexpr u32 ONE[8] = {0x4ffffffbu, 0xac96341cu, 0x9f60cd29u, 0x36fc7695u, 0x7879462eu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
  static constexpr u32 R2[8] = {0xae216da7u, 0x1bb8e645u, 0xe35c59e3u, 0x53fe3ab1u, 0x53bb8085u, 0x8c49833du, 0x7f4e44a5u, 0x0216d0b1u};
};
struct Bn254FqParams {
  static constexpr int N = 8;
  static constexpr u32 MOD[8] = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
  static constexpr u32 INV = 0xe4866389u;
  static constexpr u32 ONE[8] = {0xc58f0d9du, 0xd35d438du, 0xf5c70b3du, 0x0a78eb28u, 0x7879462cu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
  static constexpr u32 R2[8] = {0x538afa89u, 0xf32cfc5bu, 0xd44501fbu, 0xb5e71911u, 0x0a417ff6u, 0x47ab1effu, 0xcab8351fu, 0x06d89f71u};
};

// ---------------------------------------------------------------------------------
// generic limb helpers
// ---------------------------------------------------------------------------------
template <class P, int N>
__device__ __forceinline__ void ff_final_sub(u32* __restrict__ r, const u32* t, u32 top) {
  // r = (top:t) >= p ? (top:t) - p : t      (top:t < 2p guaranteed by Montgomery)
  u32 d[N];
  u32 borrow = 0;
#pragma unroll
  for (int i = 0; i < N; i++) {
    u64 x = (u64)t[i] - P::MOD[i] - borrow;
    d[i] = (u32)x;
    borrow = (u32)(x >> 63);
  }
  bool ge = (top != 0) || (borrow == 0);
#pragma unroll
  for (int i = 0; i < N; i++) r[i] = ge ? d[i] : t[i];
}

template <int N>
struct MontMulImpl;
#include "mont_mul_gen.inc"

template <class P>
struct Fp {
  static constexpr int N = P::N;
  u32 v[P::N];

  static __device__ __forceinline__ Fp zero() {
    Fp r;
#pragma unroll
    for (int i = 0; i < N; i++) r.v[i] = 0;
    return r;
  }
  static __device__ __forceinline__ Fp one() {
    Fp r;
#pragma unroll
    for (int i = 0; i < N; i++) r.v[i] = P::ONE[i];
    return r;
  }
  static __device__ __forceinline__ Fp r2() {
    Fp r;
#pragma unroll
    for (int i = 0; i < N; i++) r.v[i] = P::R2[i];
    return r;
  }
  __device__ __forceinline__ bool is_zero() const {
    u32 o = 0;
#pragma unroll
    for (int i = 0; i < N; i++) o |= v[i];
    return o == 0;
  }
  __device__ __forceinline__ bool operator==(const Fp& b) const {
    u32 o = 0;
#pragma unroll
    for (int i = 0; i < N; i++) o |= (v[i] ^ b.v[i]);
    return o == 0;
  }
  __device__ __forceinline__ bool operator!=(const Fp& b) const { return !(*this == b); }
};

template <class P>
__device__ __forceinline__ Fp<P> ff_add(const Fp<P>& a, const Fp<P>& b) {
  constexpr int N = P::N;
  u32 t[N];
  u32 c = 0;
#pragma unroll
  for (int i = 0; i < N; i++) {
    u64 s = (u64)a.v[i] + b.v[i] + c;
    t[i] = (u32)s;
    c = (u32)(s >> 32);
  }
  Fp<P> r;
  ff_final_sub<P, N>(r.v, t, c);
  return r;
}

template <class P>
__device__ __forceinline__ Fp<P> ff_sub(const Fp<P>& a, const Fp<P>& b) {
  constexpr int N = P::N;
  u32 t[N];
  u32 borrow = 0;
#pragma unroll
  for (int i = 0; i < N; i++) {
    u64 x = (u64)a.v[i] - b.v[i] - borrow;
    t[i] = (u32)x;
    borrow = (u32)(x >> 63);
  }
  // if borrow: add p back
  u32 mask = 0u - borrow;
  u32 c = 0;
  Fp<P> r;
#pragma unroll
  for (int i = 0; i < N; i++) {
    u64 s = (u64)t[i] + (P::MOD[i] & mask) + c;
    r.v[i] = (u32)s;
    c = (u32)(s >> 32);
  }
  return r;
}

template <class P>
__device__ __forceinline__ Fp<P> ff_neg(const Fp<P>& a) {
  return a.is_zero() ? a : ff_sub(Fp<P>::zero(), a);
}

template <class P>
__device__ __forceinline__ Fp<P> ff_dbl(const Fp<P>& a) {
  return ff_add(a, a);
}

template <class P>
__device__ __forceinline__ Fp<P> ff_mul(const Fp<P>& a, const Fp<P>& b) {
  Fp<P> r;
  MontMulImpl<P::N>::template mul<P>(r.v, a.v, b.v);
  return r;
}

template <class P>
__device__ __forceinline__ Fp<P> ff_sqr(const Fp<P>& a) {
  return ff_mul(a, a);
}

// Montgomery -> canonical (arkworks `into_repr`): multiply by 1.
template <class P>
__device__ __forceinline__ Fp<P> ff_from_mont(const Fp<P>& a) {
  Fp<P> o = Fp<P>::zero();
  o.v[0] = 1;
  return ff_mul(a, o);
}

// canonical -> Montgomery: multiply by R^2.
template <class P>
__device__ __forceinline__ Fp<P> ff_to_mont(const Fp<P>& a) {
  return ff_mul(a, Fp<P>::r2());
}

template <class P>
__device__ __forceinline__ Fp<P> ff_pow(Fp<P> base, u64 e) {
  Fp<P> acc = Fp<P>::one();
  while (e) {
    if (e & 1) acc = ff_mul(acc, base);
    base = ff_sqr(base);
    e >>= 1;
  }
  return acc;
}

// Fermat inversion a^(p-2); used only in low-volume paths (batch inversion's
// single inverse, affine normalisation).  0 -> 0.
template <class P>
__device__ __noinline__ Fp<P> ff_inv(const Fp<P>& a) {
  constexpr int N = P::N;
  u32 e[N];
  u32 borrow = 2;   // e = p - 2 with borrow propagation (Fr's low word is 1)
#pragma unroll
  for (int i = 0; i < N; i++) {
    u64 d = (u64)P::MOD[i] - borrow;
    e[i] = (u32)d;
    borrow = (u32)(d >> 63);
  }
  Fp<P> acc = Fp<P>::one();
  for (int i = N - 1; i >= 0; i--) {
    for (int b = 31; b >= 0; b--) {
      acc = ff_sqr(acc);
      if ((e[i] >> b) & 1) acc = ff_mul(acc, a);
    }
  }
  return acc;
}

// The curve is a build-time choice: the same sources compile into libmarlin_hip.so (BLS12-381, default) and,
// with -DMH_CURVE_BN254, into libmarlin_hip_bn254.so.
#ifdef MH_CURVE_BN254
typedef Bn254FrParams CurveFrParams;
typedef Bn254FqParams CurveFqParams;
constexpr u32 FR_TWO_ADICITY = 28;
constexpr u32 FR_SHAVE_MASK_TOP32 = 0x3fffffffu;      // REPR_SHAVE_BITS = 2
#else
typedef FrParams CurveFrParams;
typedef FqParams CurveFqParams;
constexpr u32 FR_TWO_ADICITY = 32;
constexpr u32 FR_SHAVE_MASK_TOP32 = 0x7fffffffu;      // REPR_SHAVE_BITS = 1
#endif
typedef Fp<CurveFrParams> Fr;
typedef Fp<CurveFqParams> Fq;

// 16-byte vector load/store of a field element (element arrays are 32-B / 48-B
// strided and at least 16-B aligned).
template <class P>
__device__ __forceinline__ Fp<P> ff_load(const Fp<P>* p) {
  Fp<P> r;
  const uint4* q = reinterpret_cast<const uint4*>(p);
#pragma unroll
  for (int i = 0; i < P::N / 4; i++) {
    uint4 x = q[i];
    r.v[4 * i] = x.x;
    r.v[4 * i + 1] = x.y;
    r.v[4 * i + 2] = x.z;
    r.v[4 * i + 3] = x.w;
  }
  return r;
}
template <class P>
__device__ __forceinline__ void ff_store(Fp<P>* p, const Fp<P>& a) {
  uint4* q = reinterpret_cast<uint4*>(p);
#pragma unroll
  for (int i = 0; i < P::N / 4; i++) q[i] = make_uint4(a.v[4 * i], a.v[4 * i + 1], a.v[4 * i + 2], a.v[4 * i + 3]);
}
