// Pairing on the host for the product's verifier (verify_host.h): what ark-ec's `E::product_of_pairings` decides inside
// `kzg10::check` / sonic_pc's `check_elems` under `Marlin::verify` -> `PC::check_combinations` (/root/reference
// src/lib.rs:413-423) [ark-ec / ark-poly-commit 0.3, third-party, UPSTREAM-RECALLED].  Not on the prover's hot path and
// written for clarity, not speed (a verification is two products of two or three pairings: ~0.2 s here, milliseconds in
// arkworks).  One tower shape serves both build-time curves: with xi = A + u the sextic non-residue of Fq2 = Fq[u]/(u^2+1),
//
//   Fq12 = Fq[w] / (w^12 - 2A w^6 + A^2 + 1)    (w^6 = xi, so u = w^6 - A; Fq2 embeds as a + b u)
//
//   BLS12-381: A = 1; G2 = E'(Fq2): y^2 = x^3 + 4 xi (M-type twist), untwisted by (x, y) -> (x / w^2, y / w^3);
//              e(P, Q) = f_{|x|, Q}(P) ^ ((p^12 - 1) / r); a line with slope m through T, evaluated at P, is
//              -y_P + (m x_P) w^-1 + (y_T - m x_T) w^-3
//   BN254:     A = 9; G2 = E'(Fq2): y^2 = x^3 + 3 / xi (D-type twist), untwisted by (x, y) -> (x w^2, y w^3);
//              e(P, Q) = f_{t-1, Q}(P) ^ ((p^12 - 1) / r) with t - 1 = 6 x^2 -- the plain ate pairing (ark-ec's `Bn` runs
//              the shorter optimal-ate loop 6x + 2 plus two Frobenius steps); the line is y_P - (m x_P) w + (m x_T - y_T) w^3
//
// T stays on the twist, so every slope is one Fq2 division.  Any non-degenerate bilinear map G1 x G2 -> mu_r decides the
// KZG equation alike; the tests pin bilinearity, non-degeneracy and the decisions of oracle/pairing.py, which evaluates
// the same maps with T carried in E(Fq12).
#pragma once
#include "host_ff.h"

namespace hostpair {
using hostff::HFq;
using hostff::HG1Affine;
#include "pairing_consts.inc"
#ifdef MH_CURVE_BN254
constexpr uint64_t XI_A = 9;
constexpr bool TWIST_M = false;
#else
constexpr uint64_t XI_A = 1;
constexpr bool TWIST_M = true;
#endif

struct F2 { HFq a, b; };       // a + b u
inline F2 f2_add(const F2& x, const F2& y) { return {x.a + y.a, x.b + y.b}; }
inline F2 f2_sub(const F2& x, const F2& y) { return {x.a - y.a, x.b - y.b}; }
inline F2 f2_neg(const F2& x) { return {x.a.neg(), x.b.neg()}; }
inline F2 f2_mul(const F2& x, const F2& y) { return {x.a * y.a - x.b * y.b, x.a * y.b + x.b * y.a}; }
inline F2 f2_scale(const F2& x, const HFq& k) { return {x.a * k, x.b * k}; }
inline bool f2_is_zero(const F2& x) { return x.a.is_zero() && x.b.is_zero(); }
inline bool f2_eq(const F2& x, const F2& y) { return x.a == y.a && x.b == y.b; }
inline F2 f2_inv(const F2& x) {
  const HFq n = (x.a * x.a + x.b * x.b).inv();
  return {x.a * n, (x.b * n).neg()};
}

struct G2Aff { F2 x, y; bool inf; };
inline const F2& g2_curve_b() {                       // b xi (M-type) or b / xi (D-type)
  static const F2 v = [] {
    const F2 xi{HFq::from_u64(XI_A), HFq::one()};
    const F2 b{HFq::from_u64(hostff::G1_B), HFq::zero()};
    return TWIST_M ? f2_mul(b, xi) : f2_mul(b, f2_inv(xi));
  }();
  return v;
}
inline bool g2_on_curve(const G2Aff& p) {
  if (p.inf) return true;
  return f2_eq(f2_mul(p.y, p.y), f2_add(f2_mul(f2_mul(p.x, p.x), p.x), g2_curve_b()));
}
// p + q on the twist; *slope receives the slope of the chord / tangent (undefined when the sum is the identity)
inline G2Aff g2_add(const G2Aff& p, const G2Aff& q, F2* slope = nullptr) {
  if (p.inf) return q;
  if (q.inf) return p;
  F2 m;
  if (f2_eq(p.x, q.x)) {
    if (!f2_eq(p.y, q.y) || f2_is_zero(p.y)) { G2Aff r; r.inf = true; r.x = r.y = F2{HFq::zero(), HFq::zero()}; return r; }
    const F2 xx = f2_mul(p.x, p.x);
    m = f2_mul(f2_add(f2_add(xx, xx), xx), f2_inv(f2_add(p.y, p.y)));
  } else {
    m = f2_mul(f2_sub(q.y, p.y), f2_inv(f2_sub(q.x, p.x)));
  }
  if (slope) *slope = m;
  G2Aff r;
  r.inf = false;
  r.x = f2_sub(f2_sub(f2_mul(m, m), p.x), q.x);
  r.y = f2_sub(f2_mul(m, f2_sub(p.x, r.x)), p.y);
  return r;
}
inline G2Aff g2_neg(const G2Aff& p) { G2Aff r = p; if (!p.inf) r.y = f2_neg(p.y); return r; }
inline G2Aff g2_mul(const G2Aff& p, const uint64_t* k, int nlimbs) {   // canonical scalar
  G2Aff acc; acc.inf = true; acc.x = acc.y = F2{HFq::zero(), HFq::zero()};
  for (int i = nlimbs - 1; i >= 0; i--)
    for (int b = 63; b >= 0; b--) { acc = g2_add(acc, acc); if ((k[i] >> b) & 1) acc = g2_add(acc, p); }
  return acc;
}

struct F12 { HFq c[12]; };     // sum c[i] w^i
inline F12 f12_zero() { F12 r; for (auto& x : r.c) x = HFq::zero(); return r; }
inline F12 f12_one() { F12 r = f12_zero(); r.c[0] = HFq::one(); return r; }
inline bool f12_eq(const F12& a, const F12& b) { for (int i = 0; i < 12; i++) if (!(a.c[i] == b.c[i])) return false; return true; }
inline F12 f12_mul(const F12& a, const F12& b) {
  HFq t[23];
  for (auto& x : t) x = HFq::zero();
  for (int i = 0; i < 12; i++) {
    if (a.c[i].is_zero()) continue;
    for (int j = 0; j < 12; j++) t[i + j] = t[i + j] + a.c[i] * b.c[j];
  }
  static const HFq red6 = HFq::from_u64(2 * XI_A), red0 = HFq::from_u64(XI_A * XI_A + 1);
  for (int k = 22; k >= 12; k--) {                 // w^12 = 2A w^6 - (A^2 + 1)
    if (t[k].is_zero()) continue;
    t[k - 6] = t[k - 6] + red6 * t[k];
    t[k - 12] = t[k - 12] - red0 * t[k];
  }
  F12 r;
  for (int i = 0; i < 12; i++) r.c[i] = t[i];
  return r;
}
inline F12 f12_from_f2(const F2& x, int shift = 0) {   // (a + b u) w^shift with u = w^6 - A, shift < 6
  static const HFq A = HFq::from_u64(XI_A);
  F12 r = f12_zero();
  r.c[shift] = x.a - A * x.b;
  r.c[6 + shift] = x.b;
  return r;
}
// w^-1 = (2A w^5 - w^11) / (A^2 + 1)  (from w (w^11 - 2A w^5) = -(A^2 + 1))
inline const F12& f12_winv() {
  static const F12 v = [] {
    const HFq d = HFq::from_u64(XI_A * XI_A + 1).inv();
    F12 r = f12_zero();
    r.c[5] = HFq::from_u64(2 * XI_A) * d;
    r.c[11] = d.neg();
    return r;
  }();
  return v;
}
inline const F12& f12_w3inv() {
  static const F12 v = f12_mul(f12_mul(f12_winv(), f12_winv()), f12_winv());
  return v;
}
// the line with slope m through T (both on the twist), evaluated at P in E(Fq)
inline F12 line_at(const F2& m, const G2Aff& T, const HG1Affine& P) {
  if (!TWIST_M) {                                  // y_P - (m x_P) w + (m x_T - y_T) w^3
    F12 r = f12_from_f2(f2_neg(f2_scale(m, P.x)), 1);
    const F12 c3 = f12_from_f2(f2_sub(f2_mul(m, T.x), T.y), 3);
    for (int i = 0; i < 12; i++) r.c[i] = r.c[i] + c3.c[i];
    r.c[0] = r.c[0] + P.y;
    return r;
  }
  F12 r = f12_mul(f12_from_f2(f2_scale(m, P.x)), f12_winv());
  const F12 c3 = f12_mul(f12_from_f2(f2_sub(T.y, f2_mul(m, T.x))), f12_w3inv());
  for (int i = 0; i < 12; i++) r.c[i] = r.c[i] + c3.c[i];
  r.c[0] = r.c[0] - P.y;
  return r;
}
// f_{loop, Q}(P); 1 when either argument is the identity
inline F12 miller_loop(const HG1Affine& P, const G2Aff& Q) {
  if (P.inf || Q.inf) return f12_one();
  G2Aff T = Q;
  F12 f = f12_one();
  auto bit = [](int i) { return (PAIRING_ATE_LOOP[i / 64] >> (i % 64)) & 1; };
  int top = 64 * PAIRING_ATE_LOOP_LIMBS - 1;
  while (!bit(top)) top--;
  for (int i = top - 1; i >= 0; i--) {
    F2 m;
    const G2Aff T2 = g2_add(T, T, &m);
    f = f12_mul(f12_mul(f, f), line_at(m, T, P));
    T = T2;
    if (bit(i)) {
      const G2Aff T3 = g2_add(T, Q, &m);
      f = f12_mul(f, line_at(m, T, P));
      T = T3;
    }
  }
  return f;
}
inline F12 final_exponentiation(const F12& f) {
  F12 acc = f12_one();
  bool started = false;
  for (int i = PAIRING_FINAL_EXP_LIMBS - 1; i >= 0; i--)
    for (int b = 63; b >= 0; b--) {
      if (started) acc = f12_mul(acc, acc);
      if ((PAIRING_FINAL_EXP[i] >> b) & 1) { acc = started ? f12_mul(acc, f) : f; started = true; }
    }
  return acc;
}
inline bool g2_in_subgroup(const G2Aff& p) { return g2_on_curve(p) && g2_mul(p, PAIRING_GROUP_ORDER, PAIRING_GROUP_ORDER_LIMBS).inf; }
inline F12 pairing(const HG1Affine& P, const G2Aff& Q) { return final_exponentiation(miller_loop(P, Q)); }
// prod_i e(P_i, Q_i) == 1 with one shared final exponentiation (ark-ec product_of_pairings)
inline bool pairing_product_is_one(const HG1Affine* Ps, const G2Aff* Qs, size_t n) {
  F12 f = f12_one();
  for (size_t i = 0; i < n; i++) f = f12_mul(f, miller_loop(Ps[i], Qs[i]));
  return f12_eq(final_exponentiation(f), f12_one());
}

}  // namespace hostpair
