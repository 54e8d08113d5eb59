// Marlin::verify on the host (no GPU involved): /root/reference src/lib.rs:315-433 for
// Marlin<Fr, PC, SimpleHashFiatShamirRng<Blake2s, ChaChaRng>> with PC = MarlinKZG10<E> or SonicKZG10<E>, E the build-time
// curve (BLS12-381 or BN254).
//
// Replays the Fiat-Shamir transcript (lib.rs:335-383), builds the query set and the linear combinations
// (src/ahp/verifier.rs:103-188, src/ahp/mod.rs:110-221), and decides `PC::check_combinations` (lib.rs:413-423) the way
// ark-poly-commit 0.3 does [third-party, UPSTREAM-RECALLED; SURVEY.md Appendix B-4, B-5]:
//   marlin_pc: per query point the LC commitments are combined with the opening-challenge powers -- one power per LC and
//     one more per degree-bounded LC, whose shifted commitment enters as shifted_comm - [value] shift_power -- and
//         e(C - [v] G - [random_v] gamma_G,  H)  ==  e(W,  beta_H - [z] H)
//   sonic_pc: one power per LC; a degree-bounded commitment is a commitment against the powers shifted by D - d, so the
//     combination is kept per degree bound and each bounded part is paired with [beta^-(D - d)] H (`check_elems`):
//         e(C_unbounded - [v] G - [random_v] gamma_G, H) * prod_d e(C_d, [beta^-(D-d)] H)  ==  e(W,  beta_H - [z] H)
// with the pairing of pairing_host.h.  Same steps as oracle/marlin.py `verify(use_pairing=True)`; the verifier key material
// is what kzg10::VerifierKey / marlin_pc::VerifierKey / sonic_pc::VerifierKey hold.
#pragma once
#include <vector>
#include "fs_host.h"
#include "pairing_host.h"

namespace hostverify {
using hostff::HFq; using hostff::HFr; using hostff::HG1; using hostff::HG1Affine;
using hostpair::G2Aff; using hostpair::F2;

struct VerifierKey {
  HG1Affine g, gamma_g;
  G2Aff h, beta_h;
  int pc = 0;                        // 0 MarlinKZG10, 1 SonicKZG10
  HG1Affine shift_h, shift_k;        // marlin_pc: powers_of_g[max_degree - (|H| - 2)], powers_of_g[max_degree - (|K| - 2)]
  G2Aff neg_h, neg_k;                // sonic_pc: [beta^-(max_degree - (|H| - 2))] H, [beta^-(max_degree - (|K| - 2))] H
};

inline bool read_fq(const uint8_t* p, HFq* out) {
  uint64_t c[hostff::FQ_L];
  memcpy(c, p, hostff::FQ_B);
  if (HFq::geq_mod(c)) return false;
  *out = HFq::from_canonical(c);
  return true;
}
inline bool read_fr(const uint8_t* p, HFr* out) {
  uint64_t c[4];
  memcpy(c, p, 32);
  if (HFr::geq_mod(c)) return false;
  *out = HFr::from_canonical(c);
  return true;
}
inline bool g1_on_curve(const HG1Affine& a) {
  if (a.inf) return true;
  return a.y.sqr() == a.x.sqr() * a.x + HFq::from_u64(hostff::G1_B);
}
// [r]P = O: `GroupAffine::deserialize` (what a stock arkworks verifier runs on proof bytes before Marlin::verify,
// src/data_structures.rs:100-110) rejects points outside the prime-order subgroup; BLS12-381's G1 has a cofactor, so an
// on-curve point may carry a cofactor component (BN254's G1 has cofactor 1: every curve point passes)
inline bool g1_in_subgroup(const HG1Affine& a) {
  if (a.inf || hostff::CURVE_ID == 1) return true;
  uint64_t r[4];
  for (int i = 0; i < 4; i++) r[i] = HFr::MOD_LIMB(i);
  return HG1::from_affine(a).mul(r, 4).is_identity();
}
// GroupAffine ToBytes image: x || y || infinity byte; the same acceptance rule as wire::get_g1_compressed
inline bool read_g1(const uint8_t* p, HG1Affine* out) {
  out->inf = p[2 * hostff::FQ_B] != 0;
  if (out->inf) { out->x = HFq::zero(); out->y = HFq::zero(); return true; }
  return read_fq(p, &out->x) && read_fq(p + hostff::FQ_B, &out->y) && g1_on_curve(*out) && g1_in_subgroup(*out);
}
constexpr size_t G1_TB = 2 * hostff::FQ_B + 1;       // 97 (BLS12-381) / 65 (BN254)

inline HFr pow_u64(HFr b, uint64_t e) { return b.pow_u64(e); }
inline uint64_t next_pow2(uint64_t n) { uint64_t p = 1; while (p < n) p <<= 1; return p; }
inline HG1 g1_mul(const HG1Affine& a, const HFr& k) {
  uint64_t c[4];
  k.to_canonical(c);
  return HG1::from_affine(a).mul(c, 4);
}
inline HG1 jac_mul(const HG1& a, const HFr& k) {
  uint64_t c[4];
  k.to_canonical(c);
  return a.mul(c, 4);
}
inline G2Aff g2_mul_fr(const G2Aff& a, const HFr& k) {
  uint64_t c[4];
  k.to_canonical(c);
  return hostpair::g2_mul(a, c, 4);
}

// returns 0 and *ok; a negative value for malformed inputs
inline int marlin_verify(const uint8_t* vk_bytes, size_t vk_len, const VerifierKey& vk, const std::vector<HFr>& public_input,
                         const uint8_t* proof, size_t proof_len, bool* ok, std::string* err, const fsh::ExternalFs* ext_fs = nullptr) {
  *ok = false;
  const bool sonic = vk.pc == 1;
  // marlin_pc::Commitment: comm || has_shifted || shifted (195 / 131 bytes); sonic_pc: a bare kzg10::Commitment
  const size_t COMM_TB = sonic ? G1_TB : 2 * G1_TB + 1;
  if (vk_len != 24 + 6 * COMM_TB) { *err = "verifier key bytes: expected index_info + 6 commitments of this scheme"; return -1; }
  if (proof_len != 9 * COMM_TB + 4 * 32 + 2 * (G1_TB + 1 + 32)) { *err = "proof: not a flat proof of this scheme"; return -1; }
  uint64_t info[3];
  memcpy(info, vk_bytes, 24);
  const uint64_t num_constraints = info[1], num_non_zero = info[2];
  const uint64_t H = next_pow2(num_constraints), K = next_pow2(num_non_zero);
  // ---- commitments by label
  struct Comm { HG1Affine comm; bool has_shifted; HG1Affine shifted; };
  auto read_comm = [&](const uint8_t* p, Comm* c) {
    if (!read_g1(p, &c->comm)) return false;
    c->has_shifted = false;
    if (sonic) return true;
    c->has_shifted = p[G1_TB] != 0;
    return read_g1(p + G1_TB + 1, &c->shifted);
  };
  // INDEXER_POLYNOMIALS order (src/ahp/mod.rs:33-36), then the prover's rounds (mod.rs:38-45)
  enum { ROW, COL, A_VAL, B_VAL, C_VAL, ROW_COL, W_, Z_A, Z_B, MASK, T_, G_1, H_1, G_2, H_2, NPOLY };
  Comm cm[NPOLY];
  for (int i = 0; i < 6; i++) if (!read_comm(vk_bytes + 24 + i * COMM_TB, &cm[i])) { *err = "verifier key: bad commitment"; return -1; }
  for (int i = 0; i < 9; i++) if (!read_comm(proof + i * COMM_TB, &cm[6 + i])) { *err = "proof: bad commitment"; return -1; }
  const uint8_t* pe = proof + 9 * COMM_TB;
  HFr ev[4];                                            // g_1(beta), g_2(gamma), t(beta), z_b(beta): label order
  for (int i = 0; i < 4; i++) if (!read_fr(pe + 32 * i, &ev[i])) { *err = "proof: evaluation out of range"; return -1; }
  struct Opening { HG1Affine w; bool has_rv; HFr rv; } op[2];
  const uint8_t* po = pe + 4 * 32;
  for (int i = 0; i < 2; i++) {
    const uint8_t* q = po + i * (G1_TB + 1 + 32);
    if (!read_g1(q, &op[i].w)) { *err = "proof: bad opening"; return -1; }
    op[i].has_rv = q[G1_TB] != 0;
    if (!read_fr(q + G1_TB + 1, &op[i].rv)) { *err = "proof: random_v out of range"; return -1; }
  }
  // ---- public input: formatted = 1 || input, padded to a power of two (lib.rs:323-333)
  std::vector<HFr> x;
  x.push_back(HFr::one());
  for (auto& v : public_input) x.push_back(v);
  const uint64_t X = next_pow2(x.size());
  while (x.size() < X) x.push_back(HFr::zero());
  // ---- transcript (lib.rs:335-383)
  fsh::FiatShamirRng fs;
  fs.ext = ext_fs;
  {
    std::vector<uint8_t> b;
    const char* name = "MARLIN-2019";
    b.insert(b.end(), name, name + 11);
    b.insert(b.end(), vk_bytes, vk_bytes + vk_len);
    for (size_t i = 1; i < x.size(); i++) fsh::put_fr(b, x[i]);
    fs.initialize(b);
  }
  auto vH = [&](const HFr& z) { return pow_u64(z, H) - HFr::one(); };
  auto absorb = [&](const uint8_t* p, size_t n) { std::vector<uint8_t> b(p, p + n); fs.absorb(b); };
  absorb(proof, 4 * COMM_TB);
  HFr alpha = fs.rand_fr();
  while (vH(alpha).is_zero()) alpha = fs.rand_fr();
  const HFr eta_a = fs.rand_fr(), eta_b = fs.rand_fr(), eta_c = fs.rand_fr();
  absorb(proof + 4 * COMM_TB, 3 * COMM_TB);
  HFr beta = fs.rand_fr();
  while (vH(beta).is_zero()) beta = fs.rand_fr();
  absorb(proof + 7 * COMM_TB, 2 * COMM_TB);
  const HFr gamma = fs.rand_fr();
  absorb(pe, 4 * 32);
  const HFr xi = fs.rand_u128_as_fr();
  const HFr g1_b = ev[0], g2_g = ev[1], t_b = ev[2], zb_b = ev[3];
  // ---- linear combinations (src/ahp/mod.rs:110-221)
  const HFr vHa = vH(alpha), vHb = vH(beta);
  // unnormalized bivariate Lagrange polynomial (v_H(a) - v_H(b)) / (a - b); |H| a^(|H|-1) on the diagonal
  const HFr r_ab = (alpha == beta) ? HFr::from_u64(H) * pow_u64(alpha, H - 1) : (vHa - vHb) * (alpha - beta).inv();
  const HFr vXb = pow_u64(beta, X) - HFr::one();
  HFr x_b = HFr::zero();                                 // x_poly(beta) = sum_i L_i(beta) x_i over the input domain
  {
    HFr omega = hostff::fr_two_adic_root();
    for (uint64_t sz = X; sz < (1ull << hostff::FR_TWO_ADICITY_H); sz <<= 1) omega = omega.sqr();
    const HFr xinv = HFr::from_u64(X).inv();
    HFr wi = HFr::one();
    for (uint64_t i = 0; i < X; i++) {
      const HFr d = beta - wi;
      if (d.is_zero()) { x_b = x[i]; break; }            // beta inside the domain: L_i(beta) = 1, the others 0
      x_b = x_b + vXb * xinv * wi * d.inv() * x[i];       // L_i(beta) = v_X(beta) w^i / (X (beta - w^i))
      wi = wi * omega;
    }
  }
  struct Term { HFr c; int poly; };                      // poly = -1: LCTerm::One
  const HFr one = HFr::one(), zero = HFr::zero();
  std::vector<Term> outer = {{one, MASK}, {r_ab * (eta_a + eta_c * zb_b), Z_A}, {r_ab * eta_b * zb_b, -1}, {zero - t_b * vXb, W_},
                             {zero - t_b * x_b, -1}, {zero - vHb, H_1}, {zero - beta * g1_b, -1}};
  const HFr vKg = pow_u64(gamma, K) - one;
  const HFr va = vHa * vHb;
  const HFr mult = gamma * g2_g + t_b * HFr::from_u64(K).inv();
  std::vector<Term> inner = {{eta_a * va, A_VAL}, {eta_b * va, B_VAL}, {eta_c * va, C_VAL}, {zero - beta * alpha * mult, -1},
                             {alpha * mult, ROW}, {beta * mult, COL}, {zero - mult, ROW_COL}, {zero - vKg, H_2}};
  // query set in BTreeSet order of the LC labels (verifier.rs:103-188): at beta {g_1, outer_sumcheck, t, z_b}, at gamma
  // {g_2, inner_sumcheck}; claimed evaluations: the transmitted ones, 0 for the sumchecks
  struct LC { std::vector<Term> terms; HFr claimed; int bounded; };     // bounded: polynomial index whose degree bound the LC keeps, or -1
  const LC at_beta[4] = {{{{one, G_1}}, g1_b, G_1}, {outer, zero, -1}, {{{one, T_}}, t_b, -1}, {{{one, Z_B}}, zb_b, -1}};
  const LC at_gamma[2] = {{{{one, G_2}}, g2_g, G_2}, {inner, zero, -1}};
  const HFr points[2] = {beta, gamma};
  bool all = true;
  for (int k = 0; k < 2; k++) {
    const LC* lcs = k == 0 ? at_beta : at_gamma;
    const int nl = k == 0 ? 4 : 2;
    HG1 combined = HG1::identity();                      // unbounded part (everything, for marlin_pc)
    HG1 combined_bounded = HG1::identity();              // sonic_pc: the part committed against the shifted powers
    int bound_of_point = -1;
    HFr value = zero;
    HFr ch = one;                                        // xi^counter
    for (int l = 0; l < nl; l++) {
      HFr constant = zero;
      HG1 lc_comm = HG1::identity();
      for (auto& t : lcs[l].terms) {
        if (t.poly < 0) constant = constant + t.c;
        else lc_comm = lc_comm.add(g1_mul(cm[t.poly].comm, t.c));
      }
      const HFr claimed = lcs[l].claimed - constant;     // constant terms move to the evaluation side
      if (sonic && lcs[l].bounded >= 0) {
        combined_bounded = combined_bounded.add(jac_mul(lc_comm, ch));
        bound_of_point = lcs[l].bounded;
      } else {
        combined = combined.add(jac_mul(lc_comm, ch));
      }
      value = value + claimed * ch;
      ch = ch * xi;
      if (!sonic && lcs[l].bounded >= 0) {
        const Comm& c = cm[lcs[l].bounded];
        if (!c.has_shifted) { *err = "degree-bounded commitment without a shifted part"; return -1; }
        const HG1Affine& sp = lcs[l].bounded == G_1 ? vk.shift_h : vk.shift_k;
        HG1 adj = HG1::from_affine(c.shifted).add(g1_mul(sp, claimed).neg());
        combined = combined.add(jac_mul(adj, ch));
        ch = ch * xi;
      }
    }
    HG1 lhs = combined.add(g1_mul(vk.g, value).neg());
    if (op[k].has_rv) lhs = lhs.add(g1_mul(vk.gamma_g, op[k].rv).neg());
    const G2Aff inner_g2 = hostpair::g2_add(vk.beta_h, hostpair::g2_neg(g2_mul_fr(vk.h, points[k])));
    HG1Affine pa[3] = {lhs.to_affine(), HG1::from_affine(op[k].w).neg().to_affine(), combined_bounded.to_affine()};
    G2Aff qa[3] = {vk.h, inner_g2, bound_of_point == G_1 ? vk.neg_h : vk.neg_k};
    all = hostpair::pairing_product_is_one(pa, qa, sonic && bound_of_point >= 0 ? 3 : 2) && all;
  }
  *ok = all;
  return 0;
}

}  // namespace hostverify
