// Host-only test hooks (g++, no HIP) for the host logic of the prover: field arithmetic, G1,
// Blake2s, ChaCha, Fiat-Shamir.  Built as libmarlin_hosttest.so and exercised by the CPU tests.
#include <stdint.h>
#include <string.h>
#include <vector>
#include "fs_host.h"
#include "host_ff.h"

using hostff::HFq; using hostff::HFr; using hostff::HG1; using hostff::HG1Affine;

extern "C" {
void ht_blake2s(const uint8_t* in, size_t n, uint8_t out[32]) { std::vector<uint8_t> v(in, in + n); fsh::Blake2s::digest(v, out); }
void ht_chacha_u64(const uint8_t seed[32], int rounds, size_t n, uint64_t* out) { fsh::ChaChaRng r(seed, rounds); for (size_t i = 0; i < n; i++) out[i] = r.next_u64(); }
void ht_fr_rand(const uint8_t seed[32], int rounds, size_t n, uint64_t* out_mont) { fsh::ChaChaRng r(seed, rounds); for (size_t i = 0; i < n; i++) { HFr x = fsh::fr_rand(r); memcpy(out_mont + 4 * i, x.v, 32); } }
// initialize(init) ; absorb(a1) ; 4 x rand_fr ; rand_u128
void ht_fs(const uint8_t* init, size_t n0, const uint8_t* a1, size_t n1, uint64_t* out_mont5) {
  fsh::FiatShamirRng fs; fs.initialize(std::vector<uint8_t>(init, init + n0)); fs.absorb(std::vector<uint8_t>(a1, a1 + n1));
  for (int i = 0; i < 4; i++) { HFr x = fs.rand_fr(); memcpy(out_mont5 + 4 * i, x.v, 32); }
  HFr u = fs.rand_u128_as_fr(); memcpy(out_mont5 + 16, u.v, 32);
}
void ht_fr_mul(const uint64_t* a, const uint64_t* b, uint64_t* r) { HFr x, y; memcpy(x.v, a, 32); memcpy(y.v, b, 32); HFr z = x * y; memcpy(r, z.v, 32); }
void ht_fr_inv(const uint64_t* a, uint64_t* r) { HFr x; memcpy(x.v, a, 32); HFr z = x.inv(); memcpy(r, z.v, 32); }
void ht_fq_mul(const uint64_t* a, const uint64_t* b, uint64_t* r) { HFq x, y; memcpy(x.v, a, 48); memcpy(y.v, b, 48); HFq z = x * y; memcpy(r, z.v, 48); }
// [k]P for affine P (x||y Montgomery), canonical k -> affine out + inf
void ht_g1_mul(const uint64_t* xy, const uint64_t* k, uint64_t* out_xy, int* inf) {
  HG1Affine a; memcpy(a.x.v, xy, 48); memcpy(a.y.v, xy + 6, 48); a.inf = false;
  HG1Affine r = HG1::from_affine(a).mul(k, 4).to_affine();
  memcpy(out_xy, r.x.v, 48); memcpy(out_xy + 6, r.y.v, 48); *inf = r.inf;
}
void ht_put_commitment(const uint64_t* xy, int has_shifted, const uint64_t* sxy, uint8_t* out195) {
  fsh::Commitment c; memcpy(c.comm.x.v, xy, 48); memcpy(c.comm.y.v, xy + 6, 48); c.comm.inf = false; c.has_shifted = has_shifted;
  if (has_shifted) { memcpy(c.shifted.x.v, sxy, 48); memcpy(c.shifted.y.v, sxy + 6, 48); c.shifted.inf = false; }
  std::vector<uint8_t> v; fsh::put_commitment(v, c); memcpy(out195, v.data(), v.size());
}
}
