// Fixed-base generation of a KZG10 structured reference string on device:
//   out[i] = [scale * tau^i] G,  i < n   (affine, Montgomery coordinates)
// i.e. what ark-poly-commit's KZG10::setup computes with FixedBaseMSM for
// `powers_of_g` (scale = 1) and `powers_of_gamma_g` (scale = gamma); the
// reference reaches it through Marlin::universal_setup -> PC::setup
// (/root/reference src/lib.rs:79-96).  It is not part
// of the timed prove path; it exists so that 2^22-point test/bench SRSs can be
// produced without a CPU-side group-arithmetic loop.
//
// One thread per power: scalar s_i = scale * tau^i by square-and-multiply, then
// 32 mixed additions from a host-built table T[w][d] = [d * 256^w] G, then an
// individual Fermat inversion to normalise (simple; ~2x the work of the adds).
#pragma once
#include "g1.cuh"

namespace srs {

constexpr int WINDOWS = 32;  // 8-bit windows over 256 bits
constexpr int TABLE = 256;

__global__ __launch_bounds__(128) void powers_kernel(G1Affine* __restrict__ out, const G1Affine* __restrict__ table,
                                                     Fr tau, Fr scale, u64 n, u64 first) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fr s = ff_mul(scale, ff_pow(tau, first + i));
  s = ff_from_mont(s);
  G1Xyzz acc = G1Xyzz::identity();
  for (int w = 0; w < WINDOWS; w++) {
    u32 d = (s.v[w >> 2] >> ((w & 3) * 8)) & 0xffu;
    if (d) {
      G1Affine p = g1_load_affine(table + w * TABLE + d);
      g1_madd(acc, p.x, p.y);
    }
  }
  // scale*tau^i != 0 mod r for a valid SRS; an identity result is stored as (0, 0)
  G1Affine o;
  if (acc.is_identity()) { o.x = Fq::zero(); o.y = Fq::zero(); }
  else {
    Fq zi = ff_inv(acc.zzz);                 // 1/ZZZ
    Fq zz_inv = ff_mul(zi, acc.zz);          // ZZ/ZZZ ... = 1/Z  (ZZ = Z^2, ZZZ = Z^3)
    zz_inv = ff_sqr(zz_inv);                 // 1/ZZ
    o.x = ff_mul(acc.x, zz_inv);
    o.y = ff_mul(acc.y, zi);
  }
  ff_store(&out[i].x, o.x);
  ff_store(&out[i].y, o.y);
}

}  // namespace srs
