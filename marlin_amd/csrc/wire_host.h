// Wire format of a Marlin proof: ark-serialize 0.3 `CanonicalSerialize` / `CanonicalDeserialize` of
//   Proof<F, PC>          /root/reference src/data_structures.rs:100-110 (derive: fields in declaration order)
//   ProverMsg<F>          /root/reference src/ahp/prover.rs:84-156 (as Option<Vec<F>>; Marlin's are all EmptyMessage = None)
// with PC::Commitment / PC::BatchProof / BatchLCProof from ark-poly-commit 0.3 and the point / field encodings of ark-ec /
// ark-ff 0.3 (third-party, absent from /root/reference; layouts restated):
//   Vec<T>   u64 LE length, then the items          Option<T>  one bool byte, then the value if Some
//   Fp       ceil((MODULUS_BITS + flag bits) / 8) LE bytes of the canonical value, flags in the top bits of the last byte
//   G1Affine x with SWFlags: bit 7 = "y > -y", bit 6 = infinity (identity: x = 0); 48 bytes on BLS12-381, 32 on BN254
//   marlin_pc::Commitment {comm, shifted_comm: Option}; sonic_pc commitment = kzg10::Commitment = one G1Affine
//   kzg10::Proof {w, random_v: Option<Fr>}; BatchLCProof {proof: Vec<kzg10::Proof>, evals: Option<Vec<F>> = None}
// Host only (no device work): converts between the flat ToBytes layout mh_marlin_prove emits and these bytes.
#pragma once
#include <vector>
#include "host_ff.h"

namespace wire {
using hostff::FQ_B;
using hostff::FQ_L;
using hostff::HFq;
using hostff::HFr;

constexpr size_t G1_FLAT = 2 * FQ_B + 1;      // ToBytes: x || y || infinity byte

inline bool geq(const uint64_t* a, const uint64_t* b, int n) {
  for (int i = n - 1; i >= 0; i--) { if (a[i] > b[i]) return true; if (a[i] < b[i]) return false; }
  return true;
}
// y > -y on canonical representatives (ark-ec `SWFlags::from_y_sign(self.y > -self.y)`)
inline bool y_is_positive(const uint64_t* y_canon) {
  uint64_t neg[FQ_L];
  bool zero = true;
  for (int i = 0; i < FQ_L; i++) zero = zero && y_canon[i] == 0;
  if (zero) return false;
  unsigned __int128 borrow = 0;
  for (int i = 0; i < FQ_L; i++) {
    unsigned __int128 d = (unsigned __int128)HFq::MOD_LIMB(i) - y_canon[i] - (uint64_t)borrow;
    neg[i] = (uint64_t)d; borrow = (d >> 127) & 1;
  }
  return !geq(neg, y_canon, FQ_L);       // y > p - y
}
inline void put_u64(std::vector<uint8_t>& o, uint64_t v) { for (int i = 0; i < 8; i++) o.push_back((uint8_t)(v >> (8 * i))); }

// flat G1 (x || y || inf, canonical LE) -> compressed
inline void put_g1_compressed(std::vector<uint8_t>& o, const uint8_t* flat) {
  const size_t at = o.size();
  if (flat[2 * FQ_B]) { o.insert(o.end(), FQ_B, 0); o[at + FQ_B - 1] |= 0x40; return; }
  o.insert(o.end(), flat, flat + FQ_B);
  uint64_t y[FQ_L];
  memcpy(y, flat + FQ_B, FQ_B);
  if (y_is_positive(y)) o[at + FQ_B - 1] |= 0x80;
}
// compressed -> flat; false when x is not a field element / not on the curve / the flags are inconsistent
inline bool get_g1_compressed(const uint8_t* in, std::vector<uint8_t>& flat_out) {
  uint8_t xb[FQ_B];
  memcpy(xb, in, FQ_B);
  const uint8_t flags = xb[FQ_B - 1] & 0xC0;
  xb[FQ_B - 1] &= 0x3F;
  uint64_t xc[FQ_L];
  memcpy(xc, xb, FQ_B);
  if (flags & 0x40) {
    for (int i = 0; i < FQ_L; i++) if (xc[i]) return false;
    if (flags & 0x80) return false;
    uint64_t one[FQ_L] = {1};
    flat_out.insert(flat_out.end(), FQ_B, 0);
    flat_out.insert(flat_out.end(), (const uint8_t*)one, (const uint8_t*)one + FQ_B);
    flat_out.push_back(1);
    return true;
  }
  if (HFq::geq_mod(xc)) return false;
  const HFq x = HFq::from_canonical(xc);
  const HFq y2 = x.sqr() * x + HFq::from_u64(hostff::G1_B);
  // p = 3 mod 4 on both curves: sqrt = y2^((p + 1) / 4)
  uint64_t e[FQ_L];
  {
    unsigned __int128 c = 1;
    for (int i = 0; i < FQ_L; i++) { c += HFq::MOD_LIMB(i); e[i] = (uint64_t)c; c >>= 64; }
    for (int i = 0; i < FQ_L; i++) e[i] = (e[i] >> 2) | (i + 1 < FQ_L ? e[i + 1] << 62 : 0);
  }
  HFq y = y2.pow(e, FQ_L);
  if (y.sqr() != y2) return false;
  uint64_t yc[FQ_L];
  y.to_canonical(yc);
  if (y_is_positive(yc) != ((flags & 0x80) != 0)) { y = y.neg(); y.to_canonical(yc); }
  {
    // GroupAffine::deserialize rejects points outside the prime-order subgroup: [r]P must be the identity
    hostff::HG1Affine a; a.x = x; a.y = y; a.inf = false;
    uint64_t r[4];
    for (int i = 0; i < 4; i++) r[i] = HFr::MOD_LIMB(i);
    if (!hostff::HG1::from_affine(a).mul(r, 4).is_identity()) return false;
  }
  flat_out.insert(flat_out.end(), xb, xb + FQ_B);
  flat_out.insert(flat_out.end(), (const uint8_t*)yc, (const uint8_t*)yc + FQ_B);
  flat_out.push_back(0);
  return true;
}

inline size_t flat_len(int pc) { return 9 * (pc == 1 ? G1_FLAT : 2 * G1_FLAT + 1) + 4 * 32 + 2 * (G1_FLAT + 33); }

// flat (mh_marlin_prove) -> CanonicalSerialize
inline bool serialize(const uint8_t* flat, size_t len, int pc, std::vector<uint8_t>& out) {
  if (len != flat_len(pc)) return false;
  const uint8_t* p = flat;
  static const int per_round[3] = {4, 3, 2};
  put_u64(out, 3);
  for (int r = 0; r < 3; r++) {
    put_u64(out, per_round[r]);
    for (int k = 0; k < per_round[r]; k++) {
      put_g1_compressed(out, p); p += G1_FLAT;
      if (pc == 0) {
        const uint8_t has = *p++;
        if (has > 1) return false;
        out.push_back(has);
        if (has) put_g1_compressed(out, p);
        p += G1_FLAT;
      }
    }
  }
  put_u64(out, 4);
  out.insert(out.end(), p, p + 128); p += 128;
  put_u64(out, 3); out.push_back(0); out.push_back(0); out.push_back(0);     // prover_messages: 3 x None
  put_u64(out, 2);
  for (int k = 0; k < 2; k++) {
    put_g1_compressed(out, p); p += G1_FLAT;
    const uint8_t has = *p++;
    if (has > 1) return false;
    out.push_back(has);
    if (has) out.insert(out.end(), p, p + 32);
    p += 32;
  }
  out.push_back(0);                                                            // BatchLCProof.evals = None
  return true;
}

// CanonicalDeserialize (validating) -> flat
inline bool deserialize(const uint8_t* in, size_t len, int pc, std::vector<uint8_t>& flat) {
  size_t pos = 0;
  auto need = [&](size_t n) { return pos + n <= len; };
  auto u64 = [&](uint64_t& v) { if (!need(8)) return false; v = 0; for (int i = 0; i < 8; i++) v |= (uint64_t)in[pos + i] << (8 * i); pos += 8; return true; };
  auto g1 = [&]() { if (!need(FQ_B)) return false; bool ok = get_g1_compressed(in + pos, flat); pos += FQ_B; return ok; };
  auto fr = [&]() {
    if (!need(32)) return false;
    uint64_t c[4]; memcpy(c, in + pos, 32);
    if (HFr::geq_mod(c)) return false;
    flat.insert(flat.end(), in + pos, in + pos + 32); pos += 32; return true;
  };
  static const uint64_t per_round[3] = {4, 3, 2};
  uint64_t n;
  if (!u64(n) || n != 3) return false;
  for (int r = 0; r < 3; r++) {
    if (!u64(n) || n != per_round[r]) return false;
    for (uint64_t k = 0; k < n; k++) {
      if (!g1()) return false;
      if (pc == 0) {
        if (!need(1) || in[pos] > 1) return false;
        const uint8_t has = in[pos++];
        flat.push_back(has);
        if (has) { if (!g1()) return false; }
        else { uint64_t one[FQ_L] = {1}; flat.insert(flat.end(), FQ_B, 0); flat.insert(flat.end(), (const uint8_t*)one, (const uint8_t*)one + FQ_B); flat.push_back(1); }
      }
    }
  }
  if (!u64(n) || n != 4) return false;
  for (int k = 0; k < 4; k++) if (!fr()) return false;
  if (!u64(n) || n != 3) return false;
  for (int k = 0; k < 3; k++) { if (!need(1) || in[pos] != 0) return false; pos++; }
  if (!u64(n) || n != 2) return false;
  for (int k = 0; k < 2; k++) {
    if (!g1()) return false;
    if (!need(1) || in[pos] > 1) return false;
    const uint8_t has = in[pos++];
    flat.push_back(has);
    if (has) { if (!fr()) return false; } else flat.insert(flat.end(), 32, 0);
  }
  if (!need(1) || in[pos] != 0) return false;
  pos++;
  return pos == len && flat.size() == flat_len(pc);
}

}  // namespace wire
