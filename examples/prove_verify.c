/* End to end through the C ABI alone (BLS12-381, MarlinKZG10): known-tau test SRS on the device, Marlin::index,
 * Marlin::prove, CanonicalSerialize bytes, and Marlin::verify with the host pairing -- what a binding in any language
 * (the Rust shim under shim/, cgo, JNI, ctypes) does, in plain C.
 *
 * The circuit is benches/bench.rs' DummyCircuit with a = b = 1 (so c = 1): every value is the field's one and the example
 * needs no field arithmetic of its own.  Any four limbs below r are the Montgomery form of SOME field element, which is all
 * a test SRS needs from tau and gamma.
 *
 *   gcc -std=c99 -O2 -Iinclude examples/prove_verify.c -Lmarlin_amd -lmarlin_hip -Wl,-rpath,$PWD/marlin_amd -o /tmp/prove_verify
 *   /tmp/prove_verify [log2(constraints), default 12]
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "marlin_hip.h"

#define CHECK(call)                                                                                  \
  do {                                                                                               \
    int rc_ = (call);                                                                                \
    if (rc_ != MH_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, mh_last_error()); return 1; } \
  } while (0)

static const uint64_t FR_ONE[4] = {0x00000001fffffffeull, 0x5884b7fa00034802ull, 0x998c4fefecbc4ff5ull, 0x1824b159acc5056full};
static const uint64_t TAU[4] = {0x243f6a8885a308d3ull, 0x13198a2e03707344ull, 0xa4093822299f31d0ull, 0x082efa98ec4e6c89ull};
static const uint64_t GAMMA[4] = {0x452821e638d01377ull, 0xbe5466cf34e90c6cull, 0xc0ac29b7c97c50ddull, 0x3f84d5b5b5470917ull};
/* the generator of G2 (x.c0, x.c1, y.c0, y.c1; Montgomery limbs); kzg10::setup draws h at random, any point of G2 serves */
static const uint64_t G2_GEN[24] = {
    0xf5f28fa202940a10ull, 0xb3f5fb2687b4961aull, 0xa1a893b53e2ae580ull, 0x9894999d1a3caee9ull, 0x6f67b7631863366bull, 0x058191924350bcd7ull,
    0xa5a9c0759e23f606ull, 0xaaa0c59dbccd60c3ull, 0x3bb17e18e2867806ull, 0x1b1ab6cc8541b367ull, 0xc2b6ed0ef2158547ull, 0x11922a097360edf3ull,
    0x4c730af860494c4aull, 0x597cfa1f5e369c5aull, 0xe7e6856caa0a635aull, 0xbbefb5e96e0d495full, 0x07d3a975f0ef25a2ull, 0x0083fd8e7e80dae5ull,
    0xadc0fc92df64b05dull, 0x18aa270a2b1461dcull, 0x86adac6a3be4eba0ull, 0x79495c4ec93da33aull, 0xe7175850a43ccaedull, 0x0b2bc2a163de1bf2ull};

static uint64_t np2(uint64_t n) { uint64_t p = 1; while (p < n) p <<= 1; return p; }

int main(int argc, char** argv) {
  const unsigned log_n = argc > 1 ? (unsigned)atoi(argv[1]) : 12;
  const uint64_t nc = 1ull << log_n, ni = 2, rows = nc - 1;      /* DummyCircuit: nc - 1 copies of a * b = c */
  int curve, frl, fql, adicity;
  CHECK(mh_curve_info(&curve, &frl, &fql, &adicity));
  if (curve != MH_CURVE_BLS12_381_G1) { fprintf(stderr, "this example holds BLS12-381 constants\n"); return 1; }
  CHECK(mh_init(0));

  /* ---- Marlin::universal_setup (src/lib.rs:79-96) for a known tau: powers_of_g, powers_of_gamma_g ---- */
  const uint64_t H = np2(nc), K = np2(3 * nc);
  uint64_t max_degree = 3 * H - 1;                                /* AHPForR1CS::max_degree (src/ahp/mod.rs:71-93) */
  if (K - 1 > max_degree) max_degree = K - 1;
  uint64_t srs_g, srs_gamma_g, srs_h;
  CHECK(mh_srs_powers(curve, TAU, FR_ONE, 0, max_degree + 1, &srs_g));
  CHECK(mh_srs_powers(curve, TAU, GAMMA, 0, 3, &srs_gamma_g));

  /* ---- the padded square R1CS as CSR: A picks a (column ni), B picks b (ni + 1), C picks c (column 1) ---- */
  uint64_t* row_ptr = (uint64_t*)malloc((nc + 1) * sizeof(uint64_t));
  uint32_t* col[3];
  for (uint64_t r = 0; r <= nc; r++) row_ptr[r] = r < rows ? r : rows;
  for (int k = 0; k < 3; k++) {
    col[k] = (uint32_t*)malloc(rows * sizeof(uint32_t));
    for (uint64_t e = 0; e < rows; e++) col[k][e] = k == 0 ? (uint32_t)ni : k == 1 ? (uint32_t)ni + 1 : 1u;
  }
  mh_r1cs_matrices m;
  memset(&m, 0, sizeof(m));
  m.num_constraints = nc; m.num_instance = ni;
  for (int k = 0; k < 3; k++) { m.row_ptr[k] = row_ptr; m.col[k] = col[k]; m.val[k] = NULL; }
  uint64_t pk;
  CHECK(mh_marlin_index(&m, srs_g, srs_gamma_g, &pk));

  /* ---- Marlin::prove (src/lib.rs:151-311): formatted input (1, c) and witness, all ones ---- */
  uint64_t* inst = (uint64_t*)malloc(ni * 32);
  uint64_t* wit = (uint64_t*)malloc((nc - ni) * 32);
  for (uint64_t i = 0; i < ni; i++) memcpy(inst + 4 * i, FR_ONE, 32);
  for (uint64_t i = 0; i < nc - ni; i++) memcpy(wit + 4 * i, FR_ONE, 32);
  uint8_t seed[32], proof[4096], wire[4096], back[4096];
  for (int i = 0; i < 32; i++) seed[i] = (uint8_t)i;
  size_t plen = 0, wlen = 0, blen = 0;
  CHECK(mh_marlin_prove(pk, inst, wit, seed, 20, proof, sizeof(proof), &plen));
  CHECK(mh_marlin_proof_serialize(proof, plen, 0, wire, sizeof(wire), &wlen));          /* bytes for Proof::deserialize */
  CHECK(mh_marlin_proof_deserialize(wire, wlen, 0, back, sizeof(back), &blen));
  if (blen != plen || memcmp(back, proof, plen)) { fprintf(stderr, "wire round trip differs\n"); return 1; }

  /* ---- Marlin::verify (src/lib.rs:315-433) on the host: the verifier key's group elements, then the pairing check ---- */
  uint64_t info[8];
  CHECK(mh_marlin_pk_info(pk, info));
  uint8_t vkb[2048];
  size_t vklen = 0;
  CHECK(mh_marlin_vk_bytes(pk, vkb, sizeof(vkb), &vklen));
  uint64_t g[12], gamma_g[12], shift_h[12], shift_k[12], beta_h[24];
  CHECK(mh_bases_download(srs_g, 0, 1, g));
  CHECK(mh_bases_download(srs_gamma_g, 0, 1, gamma_g));
  CHECK(mh_bases_download(srs_g, info[5] - (info[0] - 2), 1, shift_h));                  /* powers_of_g[max_degree - (|H| - 2)] */
  CHECK(mh_bases_download(srs_g, info[5] - (info[1] - 2), 1, shift_k));
  CHECK(mh_g2_srs_powers(curve, G2_GEN, TAU, FR_ONE, 1, 1, &srs_h));                      /* beta_h = [tau] h */
  CHECK(mh_g2_bases_download(srs_h, 0, 1, beta_h));
  mh_verifier_key vk = {g, gamma_g, G2_GEN, beta_h, shift_h, shift_k};
  int ok = 0, bad = 1;
  CHECK(mh_marlin_verify(vkb, vklen, &vk, 0, FR_ONE, 1, back, blen, &ok));               /* public input c = 1 */
  uint64_t wrong[4];
  memcpy(wrong, TAU, 32);
  CHECK(mh_marlin_verify(vkb, vklen, &vk, 0, wrong, 1, back, blen, &bad));
  printf("2^%u constraints: |H| = %llu, |K| = %llu, flat proof %zu bytes, wire %zu bytes, verify = %d, verify(wrong input) = %d\n",
         log_n, (unsigned long long)info[0], (unsigned long long)info[1], plen, wlen, ok, bad);

  CHECK(mh_marlin_pk_free(pk));
  CHECK(mh_g2_bases_free(srs_h));
  CHECK(mh_bases_free(srs_g));
  CHECK(mh_bases_free(srs_gamma_g));
  CHECK(mh_shutdown());
  free(row_ptr); free(inst); free(wit);
  for (int k = 0; k < 3; k++) free(col[k]);
  return ok == 1 && bad == 0 ? 0 : 2;
}
