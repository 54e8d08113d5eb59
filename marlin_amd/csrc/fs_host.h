// Host-side Fiat-Shamir transcript and RNG plumbing of Marlin::prove (product code):
//   SimpleHashFiatShamirRng<Blake2s, ChaChaRng>   /root/reference src/rng.rs:18-80
//   Fp256::rand rejection sampling, u128 -> Fr     ark-ff 0.3
//   ToBytes layouts hashed into the transcript     SURVEY.md Appendix B-6
// A few hundred bytes per absorb: stays on the host.
#pragma once
#include <stdint.h>
#include <string.h>
#include <string>
#include <vector>
#include "host_ff.h"

namespace fsh {

// ---------------------------------------------------------------- Blake2s (RFC 7693), 32-byte digest
struct Blake2s {
  uint32_t h[8];
  uint64_t t;
  uint8_t buf[64];
  size_t buflen;
  static uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
  Blake2s() {
    static const uint32_t IV[8] = {0x6A09E667, 0xBB67AE85, 0x3C6EF372, 0xA54FF53A, 0x510E527F, 0x9B05688C, 0x1F83D9AB, 0x5BE0CD19};
    memcpy(h, IV, sizeof(h));
    h[0] ^= 0x01010000 ^ 32;   // digest length 32, no key
    t = 0; buflen = 0;
  }
  void compress(const uint8_t* block, bool last) {
    static const uint32_t IV[8] = {0x6A09E667, 0xBB67AE85, 0x3C6EF372, 0xA54FF53A, 0x510E527F, 0x9B05688C, 0x1F83D9AB, 0x5BE0CD19};
    static const uint8_t S[10][16] = {
        {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
        {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
        {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
        {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
        {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0}};
    uint32_t m[16], v[16];
    for (int i = 0; i < 16; i++) memcpy(&m[i], block + 4 * i, 4);
    for (int i = 0; i < 8; i++) { v[i] = h[i]; v[i + 8] = IV[i]; }
    v[12] ^= (uint32_t)t; v[13] ^= (uint32_t)(t >> 32);
    if (last) v[14] = ~v[14];
#define B2G(a, b, c, d, x, y) \
    v[a] = v[a] + v[b] + x; v[d] = rotr(v[d] ^ v[a], 16); v[c] = v[c] + v[d]; v[b] = rotr(v[b] ^ v[c], 12); \
    v[a] = v[a] + v[b] + y; v[d] = rotr(v[d] ^ v[a], 8);  v[c] = v[c] + v[d]; v[b] = rotr(v[b] ^ v[c], 7);
    for (int r = 0; r < 10; r++) {
      const uint8_t* s = S[r];
      B2G(0, 4, 8, 12, m[s[0]], m[s[1]]) B2G(1, 5, 9, 13, m[s[2]], m[s[3]]) B2G(2, 6, 10, 14, m[s[4]], m[s[5]]) B2G(3, 7, 11, 15, m[s[6]], m[s[7]])
      B2G(0, 5, 10, 15, m[s[8]], m[s[9]]) B2G(1, 6, 11, 12, m[s[10]], m[s[11]]) B2G(2, 7, 8, 13, m[s[12]], m[s[13]]) B2G(3, 4, 9, 14, m[s[14]], m[s[15]])
    }
#undef B2G
    for (int i = 0; i < 8; i++) h[i] ^= v[i] ^ v[i + 8];
  }
  void update(const uint8_t* p, size_t n) {
    while (n) {
      if (buflen == 64) { t += 64; compress(buf, false); buflen = 0; }
      size_t k = 64 - buflen; if (k > n) k = n;
      memcpy(buf + buflen, p, k); buflen += k; p += k; n -= k;
    }
  }
  void finish(uint8_t out[32]) {
    t += buflen;
    memset(buf + buflen, 0, 64 - buflen);
    compress(buf, true);
    memcpy(out, h, 32);
  }
  static void digest(const std::vector<uint8_t>& in, uint8_t out[32]) { Blake2s b; b.update(in.data(), in.size()); b.finish(out); }
};

// ---------------------------------------------------------------- ChaCha RNG (rand_chacha BlockRng order)
struct ChaChaRng {
  // The output is one sequential stream of little-endian u32 words (block b = ChaCha(key, counter = b));
  // rand_chacha's BlockRng buffers 4 blocks at a time, which does not change the order in which words
  // are consumed, so the generator is modelled by its word position.
  uint32_t key[8];
  int rounds;
  uint64_t pos;          // words consumed so far
  uint64_t cached_block; // block held in buf
  uint32_t buf[16];
  ChaChaRng() : rounds(20), pos(0), cached_block(~0ull) { memset(key, 0, sizeof(key)); }
  ChaChaRng(const uint8_t seed[32], int rounds_) : rounds(rounds_), pos(0), cached_block(~0ull) { memcpy(key, seed, 32); }
  static uint32_t rotl(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }
  static void block(const uint32_t key[8], uint64_t counter, int rounds, uint32_t out[16]) {
    uint32_t st[16] = {0x61707865, 0x3320646e, 0x79622d32, 0x6b206574, key[0], key[1], key[2], key[3], key[4], key[5], key[6], key[7],
                       (uint32_t)counter, (uint32_t)(counter >> 32), 0, 0};
    uint32_t w[16];
    memcpy(w, st, sizeof(w));
#define CQR(a, b, c, d) \
    w[a] += w[b]; w[d] = rotl(w[d] ^ w[a], 16); w[c] += w[d]; w[b] = rotl(w[b] ^ w[c], 12); \
    w[a] += w[b]; w[d] = rotl(w[d] ^ w[a], 8);  w[c] += w[d]; w[b] = rotl(w[b] ^ w[c], 7);
    for (int r = 0; r < rounds; r += 2) {
      CQR(0, 4, 8, 12) CQR(1, 5, 9, 13) CQR(2, 6, 10, 14) CQR(3, 7, 11, 15)
      CQR(0, 5, 10, 15) CQR(1, 6, 11, 12) CQR(2, 7, 8, 13) CQR(3, 4, 9, 14)
    }
#undef CQR
    for (int i = 0; i < 16; i++) out[i] = w[i] + st[i];
  }
  uint32_t next_u32() {
    uint64_t b = pos >> 4;
    if (b != cached_block) { block(key, b, rounds, buf); cached_block = b; }
    return buf[(pos++) & 15];
  }
  uint64_t next_u64() { uint64_t lo = next_u32(); uint64_t hi = next_u32(); return lo | (hi << 32); }
  uint64_t word_pos() const { return pos; }
  void seek_words(uint64_t p) { pos = p; }
};

// Fp256::rand for BLS12-381 Fr: accepted raw limbs are the Montgomery representation.
template <class Rng>
static inline hostff::HFr fr_rand(Rng& rng) {
  for (;;) {
    hostff::HFr x;
    for (int i = 0; i < 4; i++) x.v[i] = rng.next_u64();
    x.v[3] &= hostff::FR_SHAVE_MASK_TOP64;
    if (!hostff::HFr::geq_mod(x.v)) return x;
  }
}

// ---------------------------------------------------------------- ToBytes layouts
static inline void put_fr(std::vector<uint8_t>& out, const hostff::HFr& x) {
  uint64_t c[4]; x.to_canonical(c);
  const uint8_t* p = (const uint8_t*)c; out.insert(out.end(), p, p + 32);
}
static inline void put_fq(std::vector<uint8_t>& out, const hostff::HFq& x) {
  uint64_t c[hostff::FQ_L]; x.to_canonical(c);
  const uint8_t* p = (const uint8_t*)c; out.insert(out.end(), p, p + hostff::FQ_B);
}
static inline void put_u64(std::vector<uint8_t>& out, uint64_t v) { const uint8_t* p = (const uint8_t*)&v; out.insert(out.end(), p, p + 8); }
// GroupAffine::write: x || y || infinity; the identity is (0, 1, true)
static inline void put_g1(std::vector<uint8_t>& out, const hostff::HG1Affine& a) {
  if (a.inf) { put_fq(out, hostff::HFq::zero()); put_fq(out, hostff::HFq::one()); out.push_back(1); }
  else { put_fq(out, a.x); put_fq(out, a.y); out.push_back(0); }
}
struct Commitment { hostff::HG1Affine comm; bool has_shifted; hostff::HG1Affine shifted; };
// marlin_pc::Commitment::write: comm || shifted_exists || (shifted or empty)
static inline void put_commitment(std::vector<uint8_t>& out, const Commitment& c) {
  put_g1(out, c.comm);
  out.push_back(c.has_shifted ? 1 : 0);
  if (c.has_shifted) put_g1(out, c.shifted);
  else { hostff::HG1Affine e; e.inf = true; put_g1(out, e); }
}

// ---------------------------------------------------------------- SimpleHashFiatShamirRng, or the caller's FS
// Marlin<F, PC, FS> is generic over `FS: FiatShamirRng` (/root/reference src/lib.rs:64-70; the trait, src/rng.rs:54-62, is
// RngCore + initialize + absorb).  `ext` != nullptr routes the three operations the prover and the verifier use to the
// caller's implementation (mh_marlin_prove_fs / mh_marlin_verify_fs); otherwise the reference's own instantiation
// SimpleHashFiatShamirRng<Blake2s, ChaChaRng> (src/rng.rs:18-80, src/test.rs:128-130) runs here.
struct ExternalFs {                       // field for field the C ABI's mh_fiat_shamir
  void* user;
  void (*initialize)(void* user, const uint8_t* input, size_t len);
  void (*absorb)(void* user, const uint8_t* input, size_t len);
  uint64_t (*next_u64)(void* user);
};
struct FiatShamirRng {
  uint8_t seed[32];
  ChaChaRng r;
  const ExternalFs* ext = nullptr;
  struct ExtRng { const ExternalFs* e; uint64_t next_u64() { return e->next_u64(e->user); } };
  void initialize(const std::vector<uint8_t>& input) {
    if (ext) { ext->initialize(ext->user, input.data(), input.size()); return; }
    Blake2s::digest(input, seed); r = ChaChaRng(seed, 20);
  }
  void absorb(const std::vector<uint8_t>& input) {
    if (ext) { ext->absorb(ext->user, input.data(), input.size()); return; }
    std::vector<uint8_t> b(input);
    b.insert(b.end(), seed, seed + 32);
    Blake2s::digest(b, seed);
    r = ChaChaRng(seed, 20);
  }
  hostff::HFr rand_fr() {
    if (ext) { ExtRng x{ext}; return fr_rand(x); }
    return fr_rand(r);
  }
  // u128::rand (rand 0.8 Standard: low word first) as a field element
  hostff::HFr rand_u128_as_fr() {
    uint64_t c[4] = {0, 0, 0, 0};
    if (ext) { c[0] = ext->next_u64(ext->user); c[1] = ext->next_u64(ext->user); }
    else { c[0] = r.next_u64(); c[1] = r.next_u64(); }
    return hostff::HFr::from_canonical(c);
  }
};

}  // namespace fsh
