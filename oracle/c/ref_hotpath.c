/* ref_hotpath.c -- CPU restatement (plain C, 64-bit limbs) of the arithmetic the
 * reference's hot path executes: radix-2 NTT over Fr and Pippenger MSM over G1, in the
 * algorithm class arkworks 0.3 uses.  One source, two curves: BLS12-381 by default
 * (libref_hotpath.so), BN254 with -DREF_CURVE_BN254 (libref_hotpath_bn254.so,
 * BASELINE.json configs[4]).
 *
 * ORACLE / TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): used by tests/ as the
 * checker at sizes the pure-Python oracle cannot reach, and by bench.py's
 * `cpu_baseline` leg (kind "port").  Never linked into or called from the product.
 *
 * The algorithms live in third-party crates that are absent from /root/reference
 * (ark-poly / ark-ec / ark-ff ^0.3.0, Cargo.toml:23-28); what is restated here is
 * their published behaviour as recalled in SURVEY.md Appendix B, anchored on the
 * reference's call sites:
 *   ref_ntt      <- GeneralEvaluationDomain::{fft,ifft}: src/ahp/prover.rs:326,350-351,
 *                   359,365,427,488,532-535,545,655,681  (B-1: natural order, omega =
 *                   2-adic root squared down, inverse includes n^-1)
 *   ref_msm      <- VariableBaseMSM::multi_scalar_mul via PC::commit src/lib.rs:172,193,
 *                   213 and PC::open_combinations src/lib.rs:292 (B-2: c = 3 if n < 32
 *                   else ceil(log2 n)*69/100 + 2; 2^c-1 Jacobian buckets per window,
 *                   mixed additions, running-sum reduction, windows in parallel)
 * Parity status: unpinned against arkworks bytes (no golden vectors exist upstream);
 * pinned against oracle/ *.py (naive DFT, naive MSM, known-dlog identities) by
 * tests/test_oracle_c.py.
 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef uint64_t u64;

/* ------------------------------------------------------------------ fields ---- */
#define FRN 4
#ifdef REF_CURVE_BN254
/* BN254 (ark-bn254 0.3; not a dependency of the reference; BASELINE.json configs[4]): public
 * parameters, y^2 = x^3 + 3, generator (1, 2), Fr two-adicity 28 (root = 5^((r-1)/2^28)); constants derived with Python
 * ints by the script that generated this block and re-checked by tests/test_oracle_c.py against oracle/fields.py */
#define FQN 4
#define FR_TWO_ADICITY 28
static const u64 FR_MOD[4] = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
static const u64 FR_INV = 0xc2e1f593efffffffull;
static const u64 FR_ONE[4] = {0xac96341c4ffffffbull, 0x36fc76959f60cd29ull, 0x666ea36f7879462eull, 0x0e0a77c19a07df2full};
static const u64 FR_R2[4] = {0x1bb8e645ae216da7ull, 0x53fe3ab1e35c59e3ull, 0x8c49833d53bb8085ull, 0x0216d0b17f4e44a5ull};
static const u64 FR_ROOT_CANON[4] = {0x9bd61b6e725b19f0ull, 0x402d111e41112ed4ull, 0x00e0a7eb8ef62abcull, 0x2a3c09f0a58a7e85ull};
static const u64 FQ_MOD[4] = {0x3c208c16d87cfd47ull, 0x97816a916871ca8dull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
static const u64 FQ_INV = 0x87d20782e4866389ull;
static const u64 FQ_ONE[4] = {0xd35d438dc58f0d9dull, 0x0a78eb28f5c70b3dull, 0x666ea36f7879462cull, 0x0e0a77c19a07df2full};
static const u64 FQ_R2[4] = {0xf32cfc5b538afa89ull, 0xb5e71911d44501fbull, 0x47ab1eff0a417ff6ull, 0x06d89f71cab8351full};
static const u64 G1_GX[4] = {1, 0, 0, 0};
static const u64 G1_GY[4] = {2, 0, 0, 0};
#else
#define FQN 6
#define FR_TWO_ADICITY 32
static const u64 FR_MOD[4] = {0xffffffff00000001ull, 0x53bda402fffe5bfeull, 0x3339d80809a1d805ull, 0x73eda753299d7d48ull};
static const u64 FR_INV = 0xfffffffeffffffffull;
static const u64 FR_ONE[4] = {0x00000001fffffffeull, 0x5884b7fa00034802ull, 0x998c4fefecbc4ff5ull, 0x1824b159acc5056full};
static const u64 FR_R2[4] = {0xc999e990f3f29c6dull, 0x2b6cedcb87925c23ull, 0x05d314967254398full, 0x0748d9d99f59ff11ull};
static const u64 FR_ROOT_CANON[4] = {0x3829971f439f0d2bull, 0xb63683508c2280b9ull, 0xd09b681922c813b4ull, 0x16a2a19edfe81f20ull};
static const u64 FQ_MOD[6] = {0xb9feffffffffaaabull, 0x1eabfffeb153ffffull, 0x6730d2a0f6b0f624ull,
                              0x64774b84f38512bfull, 0x4b1ba7b6434bacd7ull, 0x1a0111ea397fe69aull};
static const u64 FQ_INV = 0x89f3fffcfffcfffdull;
static const u64 FQ_ONE[6] = {0x760900000002fffdull, 0xebf4000bc40c0002ull, 0x5f48985753c758baull,
                              0x77ce585370525745ull, 0x5c071a97a256ec6dull, 0x15f65ec3fa80e493ull};
static const u64 FQ_R2[6] = {0xf4df1f341c341746ull, 0x0a76e6a609d104f1ull, 0x8de5476c4c95b6d5ull,
                             0x67eb88a9939d83c0ull, 0x9a793e85b519952dull, 0x11988fe592cae3aaull};
/* G1 generator (canonical), SURVEY Appendix D */
static const u64 G1_GX[6] = {0xfb3af00adb22c6bbull, 0x6c55e83ff97a1aefull, 0xa14e3a3f171bac58ull,
                             0xc3688c4f9774b905ull, 0x2695638c4fa9ac0full, 0x17f1d3a73197d794ull};
static const u64 G1_GY[6] = {0x0caa232946c5e7e1ull, 0xd03cc744a2888ae4ull, 0x00db18cb2c04b3edull,
                             0xfcf5e095d5d00af6ull, 0xa09e30ed741d8ae4ull, 0x08b3f481e3aaa0f1ull};

#endif
#define FQB (FQN * 8)
#define FR_BITS (FQN == 6 ? 255 : 254)
#define DEF_FIELD(P, N, MOD, INV)                                                            \
  static inline int P##_geq(const u64* a) {                                                  \
    for (int i = N - 1; i >= 0; i--) { if (a[i] > MOD[i]) return 1; if (a[i] < MOD[i]) return 0; } \
    return 1;                                                                                \
  }                                                                                          \
  static inline void P##_subm(u64* a) {                                                      \
    u64 b = 0;                                                                               \
    for (int i = 0; i < N; i++) { u128 d = (u128)a[i] - MOD[i] - b; a[i] = (u64)d; b = (u64)(d >> 127); } \
  }                                                                                          \
  static inline void P##_add(u64* r, const u64* a, const u64* b) {                           \
    u64 c = 0;                                                                               \
    for (int i = 0; i < N; i++) { u128 s = (u128)a[i] + b[i] + c; r[i] = (u64)s; c = (u64)(s >> 64); } \
    if (c || P##_geq(r)) P##_subm(r);                                                        \
  }                                                                                          \
  static inline void P##_sub(u64* r, const u64* a, const u64* b) {                           \
    u64 bo = 0;                                                                              \
    for (int i = 0; i < N; i++) { u128 d = (u128)a[i] - b[i] - bo; r[i] = (u64)d; bo = (u64)(d >> 127); } \
    if (bo) { u64 c = 0; for (int i = 0; i < N; i++) { u128 s = (u128)r[i] + MOD[i] + c; r[i] = (u64)s; c = (u64)(s >> 64); } } \
  }                                                                                          \
  static inline void P##_mul(u64* r, const u64* a, const u64* b) {                           \
    u64 t[N + 2];                                                                            \
    memset(t, 0, sizeof(t));                                                                 \
    for (int i = 0; i < N; i++) {                                                            \
      u64 c = 0;                                                                             \
      for (int j = 0; j < N; j++) { u128 p = (u128)a[i] * b[j] + t[j] + c; t[j] = (u64)p; c = (u64)(p >> 64); } \
      u128 s = (u128)t[N] + c; t[N] = (u64)s; t[N + 1] = (u64)(s >> 64);                     \
      u64 m = t[0] * INV;                                                                    \
      u128 p = (u128)m * MOD[0] + t[0]; c = (u64)(p >> 64);                                  \
      for (int j = 1; j < N; j++) { p = (u128)m * MOD[j] + t[j] + c; t[j - 1] = (u64)p; c = (u64)(p >> 64); } \
      s = (u128)t[N] + c; t[N - 1] = (u64)s; t[N] = t[N + 1] + (u64)(s >> 64);               \
    }                                                                                        \
    memcpy(r, t, N * 8);                                                                     \
    if (t[N] || P##_geq(r)) P##_subm(r);                                                     \
  }                                                                                          \
  static inline int P##_is_zero(const u64* a) { u64 o = 0; for (int i = 0; i < N; i++) o |= a[i]; return o == 0; } \
  static inline int P##_eq(const u64* a, const u64* b) { return memcmp(a, b, N * 8) == 0; }

DEF_FIELD(fr, FRN, FR_MOD, FR_INV)
DEF_FIELD(fq, FQN, FQ_MOD, FQ_INV)

static void fr_pow(u64* r, const u64* base, u64 e) {
  u64 acc[4], b[4];
  memcpy(acc, FR_ONE, 32); memcpy(b, base, 32);
  while (e) { if (e & 1) fr_mul(acc, acc, b); fr_mul(b, b, b); e >>= 1; }
  memcpy(r, acc, 32);
}
static void fr_inv(u64* r, const u64* a) {  /* a^(r-2) */
  u64 e[4]; memcpy(e, FR_MOD, 32); e[0] -= 2;
  u64 acc[4]; memcpy(acc, FR_ONE, 32);
  for (int i = 3; i >= 0; i--) for (int b = 63; b >= 0; b--) { fr_mul(acc, acc, acc); if ((e[i] >> b) & 1) fr_mul(acc, acc, a); }
  memcpy(r, acc, 32);
}
static void fq_inv(u64* r, const u64* a) {
  u64 e[FQN]; memcpy(e, FQ_MOD, FQB); e[0] -= 2;
  u64 acc[FQN]; memcpy(acc, FQ_ONE, FQB);
  for (int i = FQN - 1; i >= 0; i--) for (int b = 63; b >= 0; b--) { fq_mul(acc, acc, acc); if ((e[i] >> b) & 1) fq_mul(acc, acc, a); }
  memcpy(r, acc, FQB);
}

/* ------------------------------------------------------------------ NTT ------- */
/* Radix-2 in place, natural order in/out.  Forward: DIF butterflies then bit reversal;
 * inverse: bit reversal then DIT butterflies with omega^-1, then * n^-1 (the loop
 * structure ark-poly's Radix2EvaluationDomain uses [UPSTREAM-RECALLED B-1]). */
static void bitrev_permute(u64* a, uint32_t log_n) {
  u64 n = 1ull << log_n;
  for (u64 i = 0; i < n; i++) {
    u64 j = 0, x = i;
    for (uint32_t b = 0; b < log_n; b++) { j = (j << 1) | (x & 1); x >>= 1; }
    if (i < j) { u64 t[4]; memcpy(t, a + 4 * i, 32); memcpy(a + 4 * i, a + 4 * j, 32); memcpy(a + 4 * j, t, 32); }
  }
}

/* threads <= 1: the serial loops; otherwise the butterflies of every stage (and the twiddle table, the bit reversal, the
 * n^-1 scaling) are split over `threads` OpenMP threads -- ark-poly's `parallel` feature does the same with rayon chunks
 * (SURVEY.md 2.3 [UPSTREAM-RECALLED]); results are identical. */
int ref_ntt_mt(u64* a, uint32_t log_n, int inverse, int threads) {
  if (log_n > FR_TWO_ADICITY) return -1;
  u64 n = 1ull << log_n;
  if (n == 1) return 0;
  if (threads < 1) threads = 1;
  u64 root32[4], w_n[4];
  fr_mul(root32, FR_ROOT_CANON, FR_R2);                 /* to Montgomery */
  memcpy(w_n, root32, 32);
  for (uint32_t i = log_n; i < FR_TWO_ADICITY; i++) fr_mul(w_n, w_n, w_n);
  if (inverse) fr_inv(w_n, w_n);
  /* twiddle table w_n^k, k < n/2: chunks of 4096 started from w_n^(chunk start) */
  u64* tw = (u64*)malloc((size_t)(n / 2) * 32);
  if (!tw) return -2;
  {
    const u64 half = n / 2, CH = 4096;
    const long nch = (long)((half + CH - 1) / CH);
#pragma omp parallel for num_threads(threads) schedule(static)
    for (long c = 0; c < nch; c++) {
      u64 k0 = (u64)c * CH, k1 = k0 + CH < half ? k0 + CH : half;
      fr_pow(tw + 4 * k0, w_n, k0);
      for (u64 k = k0 + 1; k < k1; k++) fr_mul(tw + 4 * k, tw + 4 * (k - 1), w_n);
    }
  }
  const long nb = (long)(n / 2);
  if (!inverse) {
    /* DIF: gap n/2 .. 1; butterfly b = (block s, offset k) */
    for (u64 gap = n / 2; gap >= 1; gap >>= 1) {
      u64 step = (n / 2) / gap;
#pragma omp parallel for num_threads(threads) schedule(static)
      for (long b = 0; b < nb; b++) {
        u64 k = (u64)b & (gap - 1), s = ((u64)b - k) << 1;
        u64 *x = a + 4 * (s + k), *y = a + 4 * (s + k + gap), t[4];
        fr_sub(t, x, y);
        fr_add(x, x, y);
        fr_mul(y, t, tw + 4 * (k * step));
      }
    }
    bitrev_permute(a, log_n);
  } else {
    bitrev_permute(a, log_n);
    for (u64 gap = 1; gap < n; gap <<= 1) {
      u64 step = (n / 2) / gap;
#pragma omp parallel for num_threads(threads) schedule(static)
      for (long b = 0; b < nb; b++) {
        u64 k = (u64)b & (gap - 1), s = ((u64)b - k) << 1;
        u64 *x = a + 4 * (s + k), *y = a + 4 * (s + k + gap), t[4];
        fr_mul(t, y, tw + 4 * (k * step));
        fr_sub(y, x, t);
        fr_add(x, x, t);
      }
    }
    u64 nn[4] = {n, 0, 0, 0}, ninv[4];
    fr_mul(nn, nn, FR_R2);
    fr_inv(ninv, nn);
#pragma omp parallel for num_threads(threads) schedule(static)
    for (long i = 0; i < (long)n; i++) fr_mul(a + 4 * i, a + 4 * i, ninv);
  }
  free(tw);
  return 0;
}
int ref_ntt(u64* a, uint32_t log_n, int inverse) { return ref_ntt_mt(a, log_n, inverse, 1); }

/* Montgomery <-> canonical helpers for Fr vectors (arkworks into_repr / from_repr) */
void ref_fr_from_mont(u64* a, size_t n) { static const u64 one[4] = {1, 0, 0, 0}; for (size_t i = 0; i < n; i++) fr_mul(a + 4 * i, a + 4 * i, one); }
void ref_fr_to_mont(u64* a, size_t n) { for (size_t i = 0; i < n; i++) fr_mul(a + 4 * i, a + 4 * i, FR_R2); }
void ref_fr_mul_vec(u64* r, const u64* a, const u64* b, size_t n) { for (size_t i = 0; i < n; i++) fr_mul(r + 4 * i, a + 4 * i, b + 4 * i); }


/* ------------------------------------------------------------------ polynomials - */
/* Dense-polynomial helpers for the opening proofs at sizes the pure-Python oracle cannot reach (tests only): what
 * ark-poly-commit 0.3 `marlin_pc::open_combinations` / `kzg10::open` do with coefficient vectors on the way to the
 * witness MSMs (call site /root/reference src/lib.rs:292-302; SURVEY.md Appendix B-3, B-4 [UPSTREAM-RECALLED]).
 * All values Montgomery, coefficient vectors low degree first. */

/* out[i] = sum_t coef[t] * src[t][i] (i < len[t]), i < n: the LC polynomials of `open_combinations` and the
 * challenge-weighted combination sum_j xi_j p_j of `open` */
int ref_fr_lincomb(u64* out, size_t n, int nterms, const u64* const* src, const size_t* len, const u64* coef, int threads) {
  if (threads < 1) threads = 1;
#pragma omp parallel for num_threads(threads) schedule(static)
  for (long i = 0; i < (long)n; i++) {
    u64 acc[4] = {0, 0, 0, 0}, t[4];
    for (int k = 0; k < nterms; k++)
      if ((size_t)i < len[k]) { fr_mul(t, src[k] + 4 * i, coef + 4 * k); fr_add(acc, acc, t); }
    memcpy(out + 4 * i, acc, 32);
  }
  return 0;
}

/* q = (p - p(z)) / (X - z): kzg10::open's witness polynomial by synthetic division (DensePolynomial / (X - z), remainder
 * dropped); q gets len - 1 coefficients; returns p(z) in rem (may be NULL) */
void ref_fr_div_linear(u64* q, const u64* p, size_t len, const u64* z, u64* rem) {
  u64 run[4] = {0, 0, 0, 0}, t[4];
  for (size_t i = len; i-- > 0;) {
    fr_mul(t, run, z); fr_add(run, t, p + 4 * i);          /* run = p_i + z * run */
    if (i >= 1) memcpy(q + 4 * (i - 1), run, 32);
  }
  if (rem) memcpy(rem, run, 32);
}

/* p(z) by Horner (Polynomial::evaluate) */
void ref_fr_eval(const u64* p, size_t len, const u64* z, u64* out) {
  u64 run[4] = {0, 0, 0, 0}, t[4];
  for (size_t i = len; i-- > 0;) { fr_mul(t, run, z); fr_add(run, t, p + 4 * i); }
  memcpy(out, run, 32);
}

/* dst[off + i] += src[i], i < n (a shifted witness laid over `shifted_powers`' offset inside one SRS array) */
void ref_fr_add_at(u64* dst, size_t off, const u64* src, size_t n) {
  for (size_t i = 0; i < n; i++) fr_add(dst + 4 * (off + i), dst + 4 * (off + i), src + 4 * i);
}

int ref_curve_id(void) { return FQN == 6 ? 0 : 1; }

/* ------------------------------------------------------------------ G1 -------- */
typedef struct { u64 x[FQN], y[FQN], z[FQN]; } jac_t;   /* Jacobian, Z = 0 identity (arkworks GroupProjective) */

static void jac_set_identity(jac_t* p) { memcpy(p->x, FQ_ONE, FQB); memcpy(p->y, FQ_ONE, FQB); memset(p->z, 0, FQB); }
static int jac_is_identity(const jac_t* p) { return fq_is_zero(p->z); }

static void jac_double(jac_t* r, const jac_t* p) {   /* dbl-2009-l, a = 0 */
  if (jac_is_identity(p)) { *r = *p; return; }
  u64 A[FQN], B[FQN], C[FQN], D[FQN], E[FQN], F[FQN], t[FQN];
  fq_mul(A, p->x, p->x); fq_mul(B, p->y, p->y); fq_mul(C, B, B);
  fq_add(t, p->x, B); fq_mul(t, t, t); fq_sub(t, t, A); fq_sub(t, t, C); fq_add(D, t, t);
  fq_add(E, A, A); fq_add(E, E, A);
  fq_mul(F, E, E);
  u64 z3[FQN]; fq_mul(z3, p->y, p->z); fq_add(z3, z3, z3);
  u64 x3[FQN]; fq_sub(x3, F, D); fq_sub(x3, x3, D);
  u64 y3[FQN]; fq_sub(t, D, x3); fq_mul(y3, E, t);
  fq_add(C, C, C); fq_add(C, C, C); fq_add(C, C, C); fq_sub(y3, y3, C);
  memcpy(r->x, x3, FQB); memcpy(r->y, y3, FQB); memcpy(r->z, z3, FQB);
}

static void jac_add(jac_t* r, const jac_t* a, const jac_t* b) {   /* add-2007-bl */
  if (jac_is_identity(a)) { *r = *b; return; }
  if (jac_is_identity(b)) { *r = *a; return; }
  u64 z1z1[FQN], z2z2[FQN], u1[FQN], u2[FQN], s1[FQN], s2[FQN], t[FQN];
  fq_mul(z1z1, a->z, a->z); fq_mul(z2z2, b->z, b->z);
  fq_mul(u1, a->x, z2z2); fq_mul(u2, b->x, z1z1);
  fq_mul(t, b->z, z2z2); fq_mul(s1, a->y, t);
  fq_mul(t, a->z, z1z1); fq_mul(s2, b->y, t);
  if (fq_eq(u1, u2)) { if (fq_eq(s1, s2)) { jac_double(r, a); } else { jac_set_identity(r); } return; }
  u64 h[FQN], rr[FQN], hh[FQN], hhh[FQN], v[FQN];
  fq_sub(h, u2, u1); fq_sub(rr, s2, s1);
  fq_mul(hh, h, h); fq_mul(hhh, h, hh); fq_mul(v, u1, hh);
  u64 x3[FQN], y3[FQN], z3[FQN];
  fq_mul(x3, rr, rr); fq_sub(x3, x3, hhh); fq_sub(x3, x3, v); fq_sub(x3, x3, v);
  fq_sub(t, v, x3); fq_mul(y3, rr, t); fq_mul(t, s1, hhh); fq_sub(y3, y3, t);
  fq_mul(z3, a->z, b->z); fq_mul(z3, z3, h);
  memcpy(r->x, x3, FQB); memcpy(r->y, y3, FQB); memcpy(r->z, z3, FQB);
}

/* r = a + (x2, y2) affine  (add_assign_mixed) */
static void jac_add_mixed(jac_t* r, const jac_t* a, const u64* x2, const u64* y2) {
  if (jac_is_identity(a)) { memcpy(r->x, x2, FQB); memcpy(r->y, y2, FQB); memcpy(r->z, FQ_ONE, FQB); return; }
  u64 z1z1[FQN], u2[FQN], s2[FQN], t[FQN];
  fq_mul(z1z1, a->z, a->z); fq_mul(u2, x2, z1z1);
  fq_mul(t, a->z, z1z1); fq_mul(s2, y2, t);
  if (fq_eq(a->x, u2)) {
    if (fq_eq(a->y, s2)) { jac_double(r, a); } else { jac_set_identity(r); }
    return;
  }
  u64 h[FQN], rr[FQN], hh[FQN], hhh[FQN], v[FQN];
  fq_sub(h, u2, a->x); fq_sub(rr, s2, a->y);
  fq_mul(hh, h, h); fq_mul(hhh, h, hh); fq_mul(v, a->x, hh);
  u64 x3[FQN], y3[FQN], z3[FQN];
  fq_mul(x3, rr, rr); fq_sub(x3, x3, hhh); fq_sub(x3, x3, v); fq_sub(x3, x3, v);
  fq_sub(t, v, x3); fq_mul(y3, rr, t); fq_mul(t, a->y, hhh); fq_sub(y3, y3, t);
  fq_mul(z3, a->z, h);
  memcpy(r->x, x3, FQB); memcpy(r->y, y3, FQB); memcpy(r->z, z3, FQB);
}

static void jac_to_affine(const jac_t* p, u64* x, u64* y, int* inf) {
  if (jac_is_identity(p)) { memset(x, 0, FQB); memcpy(y, FQ_ONE, FQB); *inf = 1; return; }
  u64 zi[FQN], zi2[FQN], zi3[FQN];
  fq_inv(zi, p->z); fq_mul(zi2, zi, zi); fq_mul(zi3, zi2, zi);
  fq_mul(x, p->x, zi2); fq_mul(y, p->y, zi3); *inf = 0;
}

/* Jacobian X||Y||Z (18 limbs) -> affine x||y (12 limbs) + infinity flag */
void ref_g1_to_affine(const u64* xyz, u64* xy, int* inf) {
  jac_t p; memcpy(p.x, xyz, FQB); memcpy(p.y, xyz + FQN, FQB); memcpy(p.z, xyz + 2 * FQN, FQB);
  jac_to_affine(&p, xy, xy + FQN, inf);
}

static void jac_mul_canon(jac_t* r, const jac_t* p, const u64* k) {  /* k: 4 canonical limbs */
  jac_t acc; jac_set_identity(&acc);
  for (int i = 3; i >= 0; i--) for (int b = 63; b >= 0; b--) { jac_double(&acc, &acc); if ((k[i] >> b) & 1) jac_add(&acc, &acc, p); }
  *r = acc;
}

/* [k]G for a canonical scalar k -> Jacobian out (18 limbs, Montgomery coords) */
void ref_g1_mul_gen(const u64* k_canon, u64* out_xyz) {
  jac_t g; fq_mul(g.x, G1_GX, FQ_R2); fq_mul(g.y, G1_GY, FQ_R2); memcpy(g.z, FQ_ONE, FQB);
  jac_t r; jac_mul_canon(&r, &g, k_canon);
  memcpy(out_xyz, r.x, FQB); memcpy(out_xyz + FQN, r.y, FQB); memcpy(out_xyz + 2 * FQN, r.z, FQB);
}

/* bases with known discrete logs for tests: P_i = [a0 + i*d]G, i < n, affine x||y
 * Montgomery (12 limbs each); batch-normalised.  Test-data generator, not on any path
 * of the reference. */
int ref_bases_arith(const u64* a0_canon, const u64* d_canon, size_t n, u64* out_xy) {
  jac_t g; fq_mul(g.x, G1_GX, FQ_R2); fq_mul(g.y, G1_GY, FQ_R2); memcpy(g.z, FQ_ONE, FQB);
  jac_t p, d; jac_mul_canon(&p, &g, a0_canon); jac_mul_canon(&d, &g, d_canon);
  jac_t* pts = (jac_t*)malloc(n * sizeof(jac_t));
  u64* prod = (u64*)malloc(n * FQB);
  if (!pts || !prod) { free(pts); free(prod); return -2; }
  u64 acc[FQN]; memcpy(acc, FQ_ONE, FQB);
  for (size_t i = 0; i < n; i++) {
    pts[i] = p;
    if (jac_is_identity(&p)) { free(pts); free(prod); return -3; }
    fq_mul(acc, acc, p.z); memcpy(prod + FQN * i, acc, FQB);
    jac_add(&p, &p, &d);
  }
  u64 inv[FQN]; fq_inv(inv, acc);
  for (size_t i = n; i-- > 0;) {
    u64 zi[FQN];
    if (i) fq_mul(zi, inv, prod + FQN * (i - 1)); else memcpy(zi, inv, FQB);
    fq_mul(inv, inv, pts[i].z);
    u64 zi2[FQN], zi3[FQN];
    fq_mul(zi2, zi, zi); fq_mul(zi3, zi2, zi);
    fq_mul(out_xy + 2 * FQN * i, pts[i].x, zi2);
    fq_mul(out_xy + 2 * FQN * i + FQN, pts[i].y, zi3);
  }
  free(pts); free(prod);
  return 0;
}

/* KZG10::setup's powers for a KNOWN tau (test SRS; what `Marlin::universal_setup` -> `kzg10::setup` computes with its
 * FixedBaseMSM after drawing beta, /root/reference src/lib.rs:79-96): out[i] = [scale * tau^i]G, i < n, affine x||y
 * Montgomery.  Byte windows over a table T[w][d] = [d 256^w]G; batch-normalised per chunk.  tau, scale: canonical limbs. */
int ref_srs_powers(const u64* tau_canon, const u64* scale_canon, size_t n, int threads, u64* out_xy) {
  if (threads < 1) threads = 1;
  jac_t g; fq_mul(g.x, G1_GX, FQ_R2); fq_mul(g.y, G1_GY, FQ_R2); memcpy(g.z, FQ_ONE, FQB);
  jac_t* T = (jac_t*)malloc(sizeof(jac_t) * 32 * 256);
  if (!T) return -2;
  jac_t base = g;
  for (int w = 0; w < 32; w++) {
    jac_set_identity(&T[w * 256]);
    for (int d = 1; d < 256; d++) jac_add(&T[w * 256 + d], &T[w * 256 + d - 1], &base);
    jac_add(&base, &T[w * 256 + 255], &base);              /* 256 * base */
  }
  u64 tau[4], sc[4];
  memcpy(tau, tau_canon, 32); memcpy(sc, scale_canon, 32);
  fr_mul(tau, tau, FR_R2); fr_mul(sc, sc, FR_R2);           /* Montgomery */
  const size_t CH = 4096;
  const long nch = (long)((n + CH - 1) / CH);
  int rc = 0;
#pragma omp parallel for num_threads(threads) schedule(dynamic)
  for (long c = 0; c < nch; c++) {
    size_t i0 = (size_t)c * CH, i1 = i0 + CH < n ? i0 + CH : n, m = i1 - i0;
    jac_t* pts = (jac_t*)malloc(sizeof(jac_t) * m);
    u64* prod = (u64*)malloc(m * FQB);
    if (!pts || !prod) { rc = -2; free(pts); free(prod); continue; }
    u64 e[4];
    fr_pow(e, tau, (u64)i0);
    fr_mul(e, e, sc);                                       /* scale * tau^i0, Montgomery */
    u64 acc[FQN]; memcpy(acc, FQ_ONE, FQB);
    for (size_t k = 0; k < m; k++) {
      u64 can[4]; memcpy(can, e, 32);
      ref_fr_from_mont(can, 1);
      jac_t p; jac_set_identity(&p);
      for (int w = 0; w < 32; w++) {
        unsigned d = (unsigned)((can[w / 8] >> (8 * (w % 8))) & 0xff);
        if (d) jac_add(&p, &p, &T[w * 256 + d]);
      }
      pts[k] = p;
      if (jac_is_identity(&p)) { rc = -3; memcpy(prod + FQN * k, acc, FQB); }   /* scale * tau^i = 0 mod r: not an SRS */
      else { fq_mul(acc, acc, p.z); memcpy(prod + FQN * k, acc, FQB); }
      fr_mul(e, e, tau);
    }
    if (rc == 0) {
      u64 inv[FQN]; fq_inv(inv, acc);
      for (size_t k = m; k-- > 0;) {
        u64 zi[FQN];
        if (k) fq_mul(zi, inv, prod + FQN * (k - 1)); else memcpy(zi, inv, FQB);
        fq_mul(inv, inv, pts[k].z);
        u64 zi2[FQN], zi3[FQN];
        fq_mul(zi2, zi, zi); fq_mul(zi3, zi2, zi);
        fq_mul(out_xy + 2 * FQN * (i0 + k), pts[k].x, zi2);
        fq_mul(out_xy + 2 * FQN * (i0 + k) + FQN, pts[k].y, zi3);
      }
    }
    free(pts); free(prod);
  }
  free(T);
  return rc;
}

/* ------------------------------------------------------------------ MSM ------- */
typedef struct {
  const u64* bases; const u64* scalars; size_t n; unsigned c; unsigned w_start; jac_t result;
} win_job_t;

static void window_sum(win_job_t* j) {
  const unsigned c = j->c;
  const size_t nb = ((size_t)1 << c) - 1;
  jac_t* buckets = (jac_t*)malloc(nb * sizeof(jac_t));
  for (size_t b = 0; b < nb; b++) jac_set_identity(&buckets[b]);
  jac_t res; jac_set_identity(&res);
  for (size_t i = 0; i < j->n; i++) {
    const u64* s = j->scalars + 4 * i;
    if ((s[0] | s[1] | s[2] | s[3]) == 0) continue;
    /* arkworks: scalar == 1 is added directly in window 0 only */
    if (s[0] == 1 && (s[1] | s[2] | s[3]) == 0) {
      if (j->w_start == 0) jac_add_mixed(&res, &res, j->bases + 2 * FQN * i, j->bases + 2 * FQN * i + FQN);
      continue;
    }
    unsigned limb = j->w_start / 64, sh = j->w_start % 64;
    u64 d = s[limb] >> sh;
    if (sh + c > 64 && limb + 1 < 4) d |= s[limb + 1] << (64 - sh);
    d &= ((u64)1 << c) - 1;
    if (d) jac_add_mixed(&buckets[d - 1], &buckets[d - 1], j->bases + 2 * FQN * i, j->bases + 2 * FQN * i + FQN);
  }
  jac_t running; jac_set_identity(&running);
  for (size_t b = nb; b-- > 0;) { jac_add(&running, &running, &buckets[b]); jac_add(&res, &res, &running); }
  free(buckets);
  j->result = res;
}

typedef struct { win_job_t* jobs; int njobs; int next; pthread_mutex_t mu; } pool_t;
static void* worker(void* arg) {
  pool_t* p = (pool_t*)arg;
  for (;;) {
    pthread_mutex_lock(&p->mu);
    int k = p->next < p->njobs ? p->next++ : -1;
    pthread_mutex_unlock(&p->mu);
    if (k < 0) break;
    window_sum(&p->jobs[k]);
  }
  return NULL;
}

/* bases: n x (x||y) Montgomery; scalars: n x 4 limbs (Montgomery if is_mont, as arkworks holds
 * coefficients; converted with into_repr first, like KZG10::commit does); out: Jacobian X||Y||Z. */
int ref_msm(const u64* bases, const u64* scalars_in, int is_mont, size_t n, int threads, u64* out_xyz) {
  jac_t total; jac_set_identity(&total);
  if (n == 0) goto done;
  {
    u64* scalars = (u64*)malloc(n * 32);
    if (!scalars) return -2;
    memcpy(scalars, scalars_in, n * 32);
    if (is_mont) ref_fr_from_mont(scalars, n);
    unsigned c;
    if (n < 32) c = 3;
    else { unsigned lg = 0; while (((size_t)1 << lg) < n) lg++; c = lg * 69 / 100 + 2; }
    int nw = (FR_BITS + c - 1) / c;
    win_job_t* jobs = (win_job_t*)calloc(nw, sizeof(win_job_t));
    for (int w = 0; w < nw; w++) { jobs[w].bases = bases; jobs[w].scalars = scalars; jobs[w].n = n; jobs[w].c = c; jobs[w].w_start = w * c; }
    pool_t pool; pool.jobs = jobs; pool.njobs = nw; pool.next = 0; pthread_mutex_init(&pool.mu, NULL);
    if (threads < 1) threads = 1;
    if (threads > nw) threads = nw;
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * threads);
    for (int t = 1; t < threads; t++) pthread_create(&th[t], NULL, worker, &pool);
    worker(&pool);
    for (int t = 1; t < threads; t++) pthread_join(th[t], NULL);
    free(th);
    /* combine: lowest window first is special-cased upstream; mathematically Horner from the top */
    for (int w = nw - 1; w >= 0; w--) {
      for (unsigned k = 0; k < c; k++) jac_double(&total, &total);
      jac_add(&total, &total, &jobs[w].result);
    }
    free(jobs); free(scalars);
  }
done:
  memcpy(out_xyz, total.x, FQB); memcpy(out_xyz + FQN, total.y, FQB); memcpy(out_xyz + 2 * FQN, total.z, FQB);
  return 0;
}
