/* marlin_hip.h -- C ABI of libmarlin_hip.so: the MI355X (gfx950) implementation of the
 * Marlin prover hot path (MSM + NTT inside Marlin::prove and the KZG10 commit/open
 * calls underneath it).
 *
 * The reference (arkworks-rs/marlin 0.3.0, /root/reference) has no FFI seam
 * (#![forbid(unsafe_code)], src/lib.rs:16).  The entry points below are what a
 * Rust shim binds in order to drop this library in behind the reference's two
 * replaceable seams:
 *   B1  the `PC: PolynomialCommitment` type parameter of Marlin<F, PC, FS>
 *       (src/lib.rs:64,70): PC::commit (src/lib.rs:125,172,193,213) and
 *       PC::open_combinations (src/lib.rs:292) reduce to
 *       VariableBaseMSM::multi_scalar_mul  ->  mh_msm / mh_msm_dev
 *   B2  the hard-wired GeneralEvaluationDomain<F> (src/ahp/prover.rs:49-55,280-287):
 *       Radix2EvaluationDomain::{fft_in_place, ifft_in_place}  ->  mh_ntt / mh_ntt_dev
 * INTEGRATION.md shows the Rust side.
 *
 * Conventions
 *   - every function returns 0 on success, a negative MH_E* code on failure; it never
 *     throws and never aborts.  mh_last_error() returns a thread-local message.
 *   - host pointers are caller-owned.  Field elements are little-endian u64 limbs in
 *     arkworks' in-memory order and (unless stated) Montgomery form: Fr = 4 limbs,
 *     Fq = 6 limbs.  A G1 affine point is x||y (12 limbs, no infinity flag); a G1
 *     Jacobian point is X||Y||Z (18 limbs), Z = 0 meaning the identity.
 *   - a CONTEXT drives one GPU.  mh_init(device) initialises the process's default context; mh_ctx_create
 *     makes more (same or other GPUs) and mh_ctx_set_current binds the calling thread to one.  Calls into one
 *     context are serialised (its lock, its HIP stream: mh_set_stream lets the caller supply it); calls into
 *     different contexts run side by side.  Handles and device pointers belong to the context that made them.
 *   - "_dev" variants take device pointers (hipMalloc / mh_alloc / torch data_ptr).
 */
#ifndef MARLIN_HIP_H
#define MARLIN_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden: exactly what is declared in here is exported */
#pragma GCC visibility push(default)

#define MH_OK 0
#define MH_EINVAL (-1)   /* bad argument */
#define MH_ENOMEM (-2)   /* device or host allocation failed */
#define MH_EHIP (-3)     /* HIP runtime error (see mh_last_error) */
#define MH_ENOINIT (-4)  /* mh_init not called */
#define MH_ENODEV (-5)   /* no usable gfx950 device */
#define MH_ECHECK (-6)   /* an invariant of the MSM pipeline does not hold (MH_CHECK / mh_check_level): the result is not trusted */

#define MH_FIELD_BLS12_381_FR 0
#define MH_CURVE_BLS12_381_G1 0
#define MH_FIELD_BN254_FR 1
#define MH_CURVE_BN254_G1 1
/* The curve is a build-time choice: libmarlin_hip.so is BLS12-381 (Fq = 6 limbs); the same sources built with
 * -DMH_CURVE_BN254 give libmarlin_hip_bn254.so (Fr and Fq = 4 limbs, G1 affine = 8 limbs, Jacobian = 12 limbs,
 * two-adicity 28) with the identical ABI.  Each library accepts only its own curve / field id. */
int mh_curve_info(int* curve_id, int* fr_limbs64, int* fq_limbs64, int* fr_two_adicity);

/* ---- lifecycle ---------------------------------------------------------------- */
/* One process drives ONE GPU (the multi-GPU layout is one process per GPU over RCCL, see mh_marlin_set_shard), so
 * mh_init takes a single device id.  SURVEY.md 8b's sketch `mh_init(const int* device_ids, int n_devices)` is kept as
 * mh_init_devices for binding compatibility: it accepts exactly one id (n_devices == 1) and fails with MH_EINVAL
 * otherwise -- a deliberate deviation, documented in INTEGRATION.md section 2. */
int mh_init(int device_id);               /* the calling thread's current context (the default one unless mh_ctx_set_current chose another); idempotent for the same device */
int mh_init_devices(const int* device_ids, int n_devices);
int mh_shutdown(void);
const char* mh_last_error(void);
int mh_set_stream(void* hip_stream);      /* NULL -> library-owned stream */
int mh_synchronize(void);
int mh_device_info(char* name_out, size_t name_cap, int* cu_count, size_t* hbm_bytes);
/* Contexts.  The reference proves inside ONE process on rayon threads (/root/reference src/ahp/mod.rs:9-10, benches/bench.rs:1-3,
 * src/lib.rs:151-155 has one caller).  A context is one GPU's worth of library state -- streams, twiddles, workspaces, uploaded base
 * sets, prover keys, shard configuration -- so one host process can keep two proofs in flight on one GPU (two contexts, two threads)
 * or drive several GPUs (one context and one thread per GPU; mh_marlin_set_local_group joins them into one sharded prover without
 * RCCL).  mh_ctx_create initialises the new context on `device_id` (no mh_init needed); mh_ctx_set_current(ctx) binds the CALLING
 * THREAD to it (NULL: back to the default context) and every other entry point then works on it; mh_ctx_destroy releases everything
 * the context owns (no thread may be inside it). */
typedef struct mh_context* mh_ctx_t;
int mh_ctx_create(int device_id, mh_ctx_t* ctx_out);
int mh_ctx_set_current(mh_ctx_t ctx);
mh_ctx_t mh_ctx_get_current(void);
int mh_ctx_destroy(mh_ctx_t ctx);

/* ---- device memory (plain hipMalloc/hipFree/hipMemcpy on the library stream) ---- */
int mh_alloc(size_t bytes, void** dptr_out);
int mh_free(void* dptr);
int mh_memcpy_h2d(void* dst_dev, const void* src_host, size_t bytes);
int mh_memcpy_d2h(void* dst_host, const void* src_dev, size_t bytes);
int mh_memcpy_d2d(void* dst_dev, const void* src_dev, size_t bytes);
int mh_memset(void* dst_dev, int byte, size_t bytes);

/* ---- NTT over Fr: replaces ark_poly Radix2EvaluationDomain::{fft,ifft}_in_place ------
 * data: n = 2^log_n Montgomery Fr elements, natural order in and out; inverse != 0
 * also multiplies by n^-1.  log_n in [0, 32] (bounded by device memory). */
int mh_ntt(int field, uint64_t* data_mont, uint32_t log_n, int inverse);
int mh_ntt_dev(int field, const void* d_in, void* d_out, uint32_t log_n, int inverse);
/* mh_ntt for a vector that is SHORTER than the domain: ark-poly's fft_in_place / ifft_in_place first `resize(self.size(), zero)`
 * the caller's Vec and then transform it [ark-poly 0.3 radix2/mod.rs], and most forward transforms of the prover are of that kind
 * (a polynomial of H + 1 coefficients evaluated on the 4H domain, src/ahp/prover.rs:467,532-535; `const * v_H` products, :352).
 * data_mont has room for 2^log_n elements, only the first in_len are read -- and only those cross PCIe on the way in; the
 * rest of the domain counts as zero.  All 2^log_n results are written back.  in_len = 2^log_n is mh_ntt. */
int mh_ntt_len(int field, uint64_t* data_mont, size_t in_len, uint32_t log_n, int inverse);
/* Coset transforms: ark_poly Radix2EvaluationDomain::{coset_fft_in_place, coset_ifft_in_place} [ark-poly 0.3]:
 *   forward: data[i] *= g^i, then the NTT  (evaluations of the polynomial on the coset g H);
 *   inverse: the inverse NTT (with n^-1), then data[i] *= g^-i,
 * g = F::multiplicative_generator() (7 for BLS12-381 Fr, 5 for BN254 Fr).  The reference's prover calls no coset
 * transform (seam B2); BASELINE.json's north_star names it, and the patched ark-poly of shim/ routes
 * both here. */
int mh_ntt_coset(int field, uint64_t* data_mont, uint32_t log_n, int inverse);
int mh_ntt_coset_dev(int field, const void* d_in, void* d_out, uint32_t log_n, int inverse);

/* ---- MSM over G1: replaces ark_ec VariableBaseMSM::multi_scalar_mul ------------------
 * Bases are uploaded once (the SRS: KZG10 powers_of_g / powers_of_gamma_g) and
 * addressed by handle + offset, because ark-poly-commit slices one SRS array
 * (powers[num_leading_zeros..], shifted_powers[..]).  Output: Jacobian X||Y||Z. */
/* An affine point carries no infinity flag here, so the identity cannot be a base (arkworks' VariableBaseMSM accepts
 * it): both upload calls check every point against the curve equation on the device and return MH_EINVAL for a point
 * that is not on the curve -- which rejects (0, 0) / (0, 1) encodings of the identity instead of multiplying them. */
int mh_bases_upload(int curve, const uint64_t* xy_mont, size_t n, uint64_t* handle_out);
int mh_bases_from_dev(int curve, const void* d_xy_mont, size_t n, uint64_t* handle_out); /* adopts a copy */
/* Upload bases from their ark-serialize 0.3 image (what `Vec<G1Affine>` / kzg10::UniversalParams::powers_of_g look like
 * in a serialized SRS; replaces `CanonicalDeserialize for GroupAffine` on the way to Marlin::index, src/lib.rs:101-113):
 * n items back to back, WITHOUT the Vec's u64 length prefix.  compressed != 0: FQ bytes per point (x little-endian, bit 7
 * of the last byte = "y > -y", bit 6 = infinity); compressed == 0: `serialize_uncompressed`, x || y.  Decoded and
 * validated on the device (square root per point); any item that is not a finite point of the curve fails the call with
 * MH_EINVAL (SerializationError::InvalidData) -- the identity included, base sets carry no infinity flag. */
int mh_bases_upload_serialized(int curve, const uint8_t* bytes, size_t n, int compressed, uint64_t* handle_out);
/* KZG10::setup's fixed-base powers for a known-tau (test/bench) SRS:
 * bases[i] = [scale * tau^(first + i)]G, i < n, generated on the device (reference:
 * Marlin::universal_setup -> PC::setup, src/lib.rs:79-96).  `first` lets each GPU of a
 * node generate only its shard.  tau, scale: Montgomery Fr (4 limbs). */
int mh_srs_powers(int curve, const uint64_t* tau_mont, const uint64_t* scale_mont, size_t first, size_t n,
                  uint64_t* handle_out);
int mh_bases_download(uint64_t handle, size_t offset, size_t n, uint64_t* xy_mont_out);
int mh_bases_free(uint64_t handle);
int mh_bases_len(uint64_t handle, size_t* n_out);
/* Fixed-base acceleration for a base set that is multiplied again and again (the SRS): precomputes the shifts
 * 2^{start_j} * P of all window positions (W x n affine points of 128 B in device memory -- 96 B for BN254 --; W = 13 at window_bits = 20), after
 * which every MSM against this handle with enough scalars runs Pippenger over ONE shared bucket set.  window_bits
 * in [4, 20], 0 = choose from n.  Results are unchanged (an MSM has one answer).  mh_marlin_index calls this for
 * powers_of_g.  No counterpart in the reference (arkworks recomputes nothing across calls). */
int mh_bases_precompute(uint64_t handle, uint32_t window_bits);
/* window_bits / number of windows / device bytes of the handle's window table; all 0 when it has none. */
int mh_bases_table_info(uint64_t handle, uint32_t* window_bits, uint32_t* windows, uint64_t* table_bytes);
/* How many job groups (<= 8 MSMs launched together) have run on the fixed-base path / on the variable-base path
 * since mh_init: lets a caller (and the tests) see which algorithm served its MSMs. */
int mh_msm_path_counts(uint64_t* fixed_base_groups, uint64_t* variable_base_groups);
int mh_msm(uint64_t bases_handle, size_t base_offset, const uint64_t* scalars, int scalars_are_montgomery,
           size_t n, uint64_t* out_xyz_mont);
int mh_msm_dev(uint64_t bases_handle, size_t base_offset, const void* d_scalars, int scalars_are_montgomery,
               size_t n, uint64_t* out_xyz_mont);
/* Several independent MSMs through one launch sequence (the polynomials of one PC::commit call,
 * src/lib.rs:172,193,213): job j multiplies ns[j] scalars at d_scalars[j] with bases (handles[j], base_offsets[j]).
 * out_xyz: njobs x 18 limbs. */
int mh_msm_batch_dev(size_t njobs, const uint64_t* bases_handles, const size_t* base_offsets, const void* const* d_scalars,
                     const size_t* ns, int scalars_are_montgomery, uint64_t* out_xyz_mont);
/* The same with the scalar vectors in HOST memory (what a Rust host holds): the polynomials of one PC::commit call go up
 * back to back and run as one batch; a host vector that appears twice (a degree-bounded polynomial against powers and
 * shifted powers) is uploaded once. */
int mh_msm_batch(size_t njobs, const uint64_t* bases_handles, const size_t* base_offsets, const uint64_t* const* scalars,
                 const size_t* ns, int scalars_are_montgomery, uint64_t* out_xyz_mont);
/* Jacobian -> affine x||y (Montgomery) + infinity flag, on the host (GroupProjective::into_affine) */
int mh_g1_to_affine(const uint64_t* xyz_mont, uint64_t* xy_mont_out, int* is_infinity_out);

/* Sum of n Jacobian points on the host (no device needed): the combine step of a point-sharded MSM after
 * the ranks all_gather their 144-byte partial results. */
int mh_g1_sum(const uint64_t* xyz_points, size_t n, uint64_t* out_xyz);

/* ---- G2 (the verifier side of the SRS): replaces ark_ec VariableBaseMSM over G2Affine and the G2 half of
 * kzg10::setup (h, beta_h, SonicKZG10's neg_powers_of_h; reached from Marlin::universal_setup -> PC::setup,
 * src/lib.rs:79-96).  Off the prover's hot path -- Marlin::prove multiplies G1 points only -- but named by
 * BASELINE.json's north_star.  A G2 affine point is x.c0 || x.c1 || y.c0 || y.c1 (4 Fq, Montgomery, Fq2 = Fq[u]/(u^2+1)),
 * no infinity flag: uploads are checked against the twist equation.  Results are AFFINE (the library normalises on the
 * device) with an infinity flag. */
int mh_g2_bases_upload(int curve, const uint64_t* xy_mont, size_t n, uint64_t* handle_out);
/* bases[i] = [scale * tau^(first + i)] H for the caller's generator H (gen_xy, 4 Fq); scale_mont NULL = 1. */
int mh_g2_srs_powers(int curve, const uint64_t* gen_xy_mont, const uint64_t* tau_mont, const uint64_t* scale_mont, size_t first,
                     size_t n, uint64_t* handle_out);
int mh_g2_bases_download(uint64_t handle, size_t offset, size_t n, uint64_t* xy_mont_out);
int mh_g2_bases_free(uint64_t handle);
int mh_g2_msm(uint64_t handle, size_t base_offset, const uint64_t* scalars, int scalars_are_montgomery, size_t n,
              uint64_t* out_xy_mont, int* is_infinity_out);

/* ---- Marlin index / prove with device-resident polynomials --------------------------------
 * Host-side mirror of Marlin::<Fr, MarlinKZG10<Bls12_381>, SimpleHashFiatShamirRng<Blake2s,
 * ChaChaRng>>::{index, prove} (src/lib.rs:100-148, 151-311).  The R1CS is given as the padded,
 * square matrices ark-relations' `to_matrices()` yields after the reference's padding
 * (src/ahp/constraint_systems.rs:45-81): CSR per matrix, columns = instance variables first
 * (the formatted public input, leading 1 included, padded to a power of two), then witnesses. */
typedef struct {
  uint64_t num_constraints;        /* = number of variables (square) */
  uint64_t num_instance;           /* formatted public inputs incl. the leading one; power of two */
  const uint64_t* row_ptr[3];      /* A, B, C: row_ptr[k][num_constraints + 1] */
  const uint32_t* col[3];          /* column indices */
  const uint64_t* val[3];          /* Montgomery Fr coefficients, 4 limbs each; NULL = every coefficient is 1 */
} mh_r1cs_matrices;
/* srs_g: bases handle holding powers_of_g[0..=max_degree]; srs_gamma_g: >= 3 powers_of_gamma_g. */
int mh_marlin_index(const mh_r1cs_matrices* m, uint64_t srs_g, uint64_t srs_gamma_g, uint64_t* pk_out);
/* same with the polynomial-commitment scheme chosen: pc = 0 MarlinKZG10 (src/test.rs:123), pc = 1 SonicKZG10
 * (benches/bench.rs:81; needs powers_of_gamma_g[0..=max_degree+1] in srs_gamma_g).  A Sonic proof is 9 G1 commitments,
 * 4 evaluations and 2 openings (1261 bytes on BLS12-381). */
int mh_marlin_index_pc(const mh_r1cs_matrices* m, uint64_t srs_g, uint64_t srs_gamma_g, int pc, uint64_t* pk_out);
int mh_marlin_pk_free(uint64_t pk);
/* info8: |H|, |K|, |X|, num_non_zero, index max_degree, srs max_degree, num_constraints, num_instance */
int mh_marlin_pk_info(uint64_t pk, uint64_t* info8);
/* IndexVerifierKey::write bytes (index_info || 6 index commitments); out may be NULL to query the length */
int mh_marlin_vk_bytes(uint64_t pk, uint8_t* out, size_t cap, size_t* len_out);
/* instance: num_instance Montgomery Fr (formatted input); witness: num_constraints - num_instance
 * Montgomery Fr; zk_rng = rand_chacha ChaCha{8,12,20}Rng::from_seed(zk_seed).  proof_out receives
 * the flat ToBytes-layout proof (9 commitments, 4 evaluations, 2 opening proofs; 2143 bytes). */
int mh_marlin_prove(uint64_t pk, const uint64_t* instance_mont, const uint64_t* witness_mont, const uint8_t* zk_seed32,
                    int zk_chacha_rounds, uint8_t* proof_out, size_t cap, size_t* len_out);
/* The same with the formatted input and the witness already in device memory (device pointers): no PCIe transfer
 * inside the call.  bench.py times this entry point (inputs resident in HBM); mh_marlin_prove adds 32 B per
 * constraint of host-to-device copy. */
int mh_marlin_prove_dev(uint64_t pk, const void* d_instance_mont, const void* d_witness_mont, const uint8_t* zk_seed32,
                        int zk_chacha_rounds, uint8_t* proof_out, size_t cap, size_t* len_out);

/* Marlin::prove is generic over the caller's `zk_rng: &mut R` (src/lib.rs:151-155); the two entry points above replay
 * rand_chacha generators.  For any other RngCore the caller draws the field elements itself -- `Fr::rand(zk_rng)`,
 * mh_marlin_zk_draw_count(pk) of them, in the order the reference consumes them (r_w, r_za, r_zb,
 * the 3|H| mask coefficients, then 3 per hiding commitment) -- and passes them here (host memory, Montgomery limbs,
 * 4 per element).  Same proof bytes as the reference with that rng. */
int mh_marlin_zk_draw_count(uint64_t pk, size_t* n_out);
int mh_marlin_prove_draws(uint64_t pk, const uint64_t* instance_mont, const uint64_t* witness_mont, const uint64_t* zk_draws_mont,
                          size_t n_draws, uint8_t* proof_out, size_t cap, size_t* len_out);

/* Marlin<F, PC, FS> is generic over `FS: FiatShamirRng` (src/lib.rs:64-70; the trait -- RngCore + initialize + absorb -- is
 * src/rng.rs:54-62).  The entry points above run the reference's own instantiation SimpleHashFiatShamirRng<Blake2s, ChaChaRng>
 * (src/test.rs:128-130) inside the library; these take the caller's FS as three callbacks and route every transcript operation
 * of prove / verify to them: initialize(b"MARLIN-2019" || vk || public input) (src/lib.rs:161-163), absorb(round commitments,
 * then the evaluations) (:180,:201,:221,:289), and next_u64 under `F::rand(fs_rng)` -- four words per attempt, top bits shaved,
 * rejected when >= r, like Fp256::rand -- and `u128::rand(fs_rng)` (two words, low first).  Callbacks run on the calling thread,
 * between device batches; they must not call back into this library. */
typedef struct {
  void* user;
  void (*initialize)(void* user, const uint8_t* input, size_t len);
  void (*absorb)(void* user, const uint8_t* input, size_t len);
  uint64_t (*next_u64)(void* user);
} mh_fiat_shamir;
int mh_marlin_prove_fs(uint64_t pk, const uint64_t* instance_mont, const uint64_t* witness_mont, const uint8_t* zk_seed32,
                       int zk_chacha_rounds, const mh_fiat_shamir* fs, uint8_t* proof_out, size_t cap, size_t* len_out);

/* Wire format (host only, no device needed): the flat ToBytes-layout proof of mh_marlin_prove <-> the bytes of
 * ark-serialize's `CanonicalSerialize for Proof<Fr, PC>` (src/data_structures.rs:100-110; ProverMsg as Option<Vec<F>>,
 * src/ahp/prover.rs:84-99): compressed G1 (x with the y-sign / infinity flags in the top two bits of the last byte), u64
 * Vec lengths, one-byte Option tags.  pc = 0 MarlinKZG10 (855 bytes on BLS12-381), 1 SonicKZG10.  A stock arkworks
 * verifier deserialises these bytes with `Proof::deserialize`.  mh_marlin_proof_deserialize validates like
 * CanonicalDeserialize does (field range, on-curve, prime-order subgroup, flag consistency) and returns MH_EINVAL
 * otherwise.  out == NULL queries the length. */
int mh_marlin_proof_serialize(const uint8_t* flat_proof, size_t flat_len, int pc, uint8_t* out, size_t cap, size_t* len_out);
int mh_marlin_proof_deserialize(const uint8_t* bytes, size_t len, int pc, uint8_t* flat_out, size_t cap, size_t* len_out);

/* Marlin::verify (src/lib.rs:315-433) on the host (no device needed): transcript replay, linear combinations,
 * PC::check_combinations with a real pairing on the library's curve.  pc: 0 = MarlinKZG10, 1 = SonicKZG10.
 * vk_bytes = mh_marlin_vk_bytes (index_info || 6 index commitments); the group elements of kzg10::VerifierKey and
 * marlin_pc:: / sonic_pc::VerifierKey are passed separately as affine Montgomery limbs (G1: x||y, 2 * fq_limbs64;
 * G2: x.c0||x.c1||y.c0||y.c1, 4 * fq_limbs64): g, gamma_g, h, beta_h and, for the two enforced degree bounds |H| - 2 and
 * |K| - 2, the shift powers -- pc 0: G1 points powers_of_g[max_degree - bound]; pc 1: G2 points
 * [beta^-(max_degree - bound)] h (sonic_pc's degree_bounds_and_neg_powers_of_h).  G2 elements must lie in the order-r
 * subgroup.  public_input: the UNformatted input (no leading one), Montgomery Fr.  *ok_out = 1 accept / 0 reject; a
 * malformed proof or key is MH_EINVAL. */
typedef struct {
  const uint64_t* g_xy; const uint64_t* gamma_g_xy;
  const uint64_t* h_xy; const uint64_t* beta_h_xy;
  const uint64_t* shift_power_h_xy; const uint64_t* shift_power_k_xy;
} mh_verifier_key;
int mh_marlin_verify(const uint8_t* vk_bytes, size_t vk_len, const mh_verifier_key* vk, int pc, const uint64_t* public_input_mont,
                     size_t n_public, const uint8_t* flat_proof, size_t proof_len, int* ok_out);
/* the same with the caller's FS (see mh_marlin_prove_fs) */
int mh_marlin_verify_fs(const uint8_t* vk_bytes, size_t vk_len, const mh_verifier_key* vk, int pc, const uint64_t* public_input_mont,
                        size_t n_public, const uint8_t* flat_proof, size_t proof_len, const mh_fiat_shamir* fs, int* ok_out);
/* prod_i e(P_i, Q_i) == 1 (ark_ec PairingEngine::product_of_pairings followed by the comparison), host only;
 * n affine G1 points (2 * fq_limbs64 limbs each) and n affine G2 points (4 * fq_limbs64 limbs each), none of them the
 * identity; G2 points are taken to be in the order-r subgroup. */
int mh_pairing_product_is_one(const uint64_t* g1_xy_mont, const uint64_t* g2_xy_mont, size_t n, int* is_one_out);

/* Multi-GPU (one process per GPU): shard every MSM of mh_marlin_prove by BUCKET RANGE over `world` ranks.  Every rank
 * holds the full SRS, window table and prover key, runs the whole prover and recodes every scalar, but sorts,
 * accumulates and reduces only the digits that fall into ITS partitions of the shared bucket set -- partition v
 * (2^11 buckets) belongs to rank v mod world (interleaved, because low buckets are heavier).  A rank's result is a
 * partial sum; ONE all_gather per batch of MSMs exchanges the partial Jacobian points and every rank adds them, so all
 * ranks derive identical commitments and Fiat-Shamir challenges.  Groups the fixed-base path does not serve on a rank
 * (a window table with fewer partitions than ranks, short vectors, the skew fallback) are computed in full by that rank;
 * the payload carries a share/whole flag per job and a whole result takes precedence over shares.
 * The library is transport-agnostic: `allgather` must gather `bytes` bytes from every rank into recv (rank-major,
 * world * bytes) -- in this repo torch.distributed.all_gather over RCCL/xGMI (marlin_amd/dist.py).  Every rank must call
 * mh_marlin_prove with identical arguments. */
typedef int (*mh_allgather_fn)(const void* send, size_t bytes, void* recv, void* user);
int mh_marlin_set_shard(int rank, int world, mh_allgather_fn allgather, void* user);
/* mh_msm_batch_dev with the jobs sharded over the registered ranks (every rank passes the same full jobs and receives
 * the combined results): the seam-route MSM (PC::commit -> VariableBaseMSM, src/lib.rs:172,193,213,292) for a
 * multi-GPU host. */
int mh_msm_batch_sharded_dev(size_t njobs, const uint64_t* bases_handles, const size_t* base_offsets, const void* const* d_scalars,
                             const size_t* ns, int scalars_are_montgomery, uint64_t* out_xyz_mont);
int mh_marlin_probe_allgather(const void* send, size_t bytes, void* recv);  /* one all-gather through the registered transport (a caller's self-test of it) */

/* ---- one process, several GPUs ------------------------------------------------------------------------------------------------
 * The reference has ONE caller of Marlin::prove (/root/reference src/lib.rs:151-155) and parallelises inside the process (rayon,
 * src/ahp/mod.rs:9-10).  A group is `world` contexts of THIS process -- one per listed device (a device may repeat: several ranks
 * on one GPU, which is how the tests run it) -- joined by an in-process transport: host payloads meet in shared memory between
 * rendezvous of the rank threads, device buffers are pulled peer-to-peer (hipMemcpyPeerAsync over xGMI, stream-ordered behind
 * events).  No RCCL, no launcher, no second process.  mh_group_create makes and initialises the contexts and registers the
 * transport on each (as mh_marlin_set_shard / _set_alltoall / _set_allgather_dev would); mh_group_run calls fn(rank, user) on `world`
 * threads, thread r bound to context r (mh_ctx_set_current), and returns the first non-zero result: inside fn a rank uploads
 * its SRS, indexes and proves with the ordinary entry points, and every rank's proof is the one-GPU proof.  mh_group_ctx gives
 * a rank's context to a caller that runs its own threads.  A rank that does not reach a collective within MH_GROUP_TIMEOUT_S
 * (default 300) fails it on the ranks that wait. */
typedef struct mh_group* mh_group_t;
typedef int (*mh_group_fn)(int rank, void* user);
int mh_group_create(const int* device_ids, int world, mh_group_t* group_out);
int mh_group_size(mh_group_t group);
mh_ctx_t mh_group_ctx(mh_group_t group, int rank);
int mh_group_run(mh_group_t group, mh_group_fn fn, void* user);
int mh_group_destroy(mh_group_t group);

/* ---- building blocks of the slice-sharded multi-GPU pipeline (src/ahp/prover.rs:351-366,532-535,655-688 are
 * the transforms it distributes) ----
 * A polynomial of the distributed prover lives as G = world slices: coefficient vectors CYCLIC (rank r holds x[r + G j]:
 * "C-layout", independent of zero-padding), evaluation vectors on a domain of n = G m points in BLOCKS OF k mod m (rank r
 * holds X[k] for (k mod m) in [r m / G, (r + 1) m / G), locally ordered local[k1 (m / G) + t] = X[r m / G + t + m k1]:
 * "M-layout"; a point of a subdomain has the same owner in the larger domain).
 * mh_ntt_dist_dev: one transform of 2^log_n points with ONE all-to-all of 32 n / G^2 bytes per peer (4-step NTT):
 * forward takes the C-layout slice (n / G elements) and returns the M-layout block, inverse the other way round; results
 * equal mh_ntt on the gathered vector.  world must be a power of two <= 8, n >= world^2.  The all-to-all works on DEVICE
 * buffers: chunk q (bytes_per_peer bytes) of d_send goes to rank q, chunk q of d_recv comes from rank q; it must return
 * with d_recv complete (torch.distributed.all_to_all_single over RCCL in marlin_amd/dist.py). */
typedef int (*mh_alltoall_fn)(const void* d_send, size_t bytes_per_peer, void* d_recv, void* user);
/* Registering an all-to-all (next to mh_marlin_set_shard's all_gather; world = 4 or 8, |H| >= world^2) also
 * switches mh_marlin_prove to its SLICED sections: the 4H- and K-sized transforms of rounds 2 and 3 run distributed, the
 * pointwise work between them and the opening polynomials on 1 / world of each vector, with two all-gathers (this callback
 * with the same chunk for every peer) per proof -- same proof bytes.  NULL unregisters. */
int mh_marlin_set_alltoall(mh_alltoall_fn alltoall, void* user);
/* stream_ordered != 0: the registered all-to-all orders itself against the library's stream on the device (it enqueues on the
 * stream given to mh_set_stream, or makes that stream wait), so the library does not synchronise the stream before calling it and
 * the callback must not assume finished buffers on the host side.  mh_marlin_set_alltoall resets the mode to 0. */
int mh_marlin_set_alltoall_mode(int stream_ordered);
/* Optional, after mh_marlin_set_alltoall: a real all-gather of DEVICE buffers (`bytes` from every rank into d_recv, rank-major)
 * for the round polynomials of the sliced sections; without it they travel through the all-to-all, every rank sending `world`
 * copies of its chunk.  It follows the all-to-all's mode (stream-ordered or not).  NULL unregisters. */
typedef int (*mh_allgather_dev_fn)(const void* d_send, size_t bytes, void* d_recv, void* user);
int mh_marlin_set_allgather_dev(mh_allgather_dev_fn allgather_dev, void* user);
/* Runs a registered device exchange once on the caller's device buffers and waits for it: which = 0 the all-to-all (bytes per
 * peer), 1 the device all-gather.  For a transport's self-test before the first proof. */
int mh_marlin_probe_exchange_dev(int which, const void* d_send, size_t bytes, void* d_recv);

/* ---- native transport: RCCL called by the library itself (marlin_amd/csrc/rccl_native.h) --------------------------------
 * The three collectives of a sharded proof -- all-gather of partial points (src/lib.rs:172,193,213: one per PC::commit),
 * all-to-all of the distributed transforms and all-gather of round polynomials (src/ahp/prover.rs:532-535,655-688) -- issued
 * from C++ as ncclAllGather / ncclAllToAll on the library's stream: stream-ordered with the kernels around them, no
 * interpreter and no callback in the path.  librccl.so.1 is resolved with dlopen at the first call (the copy the process has
 * already mapped, if any), so a one-GPU caller never needs it.
 *   rank 0:      mh_rccl_unique_id(id)            -- ncclGetUniqueId, 128 opaque bytes
 *   the caller:  hands `id` to every rank (bench.py: one torch.distributed broadcast; a C caller: its own bootstrap)
 *   every rank:  mh_marlin_set_rccl(rank, world, id)   -- COLLECTIVE (ncclCommInitRank); afterwards mh_marlin_prove shards its MSMs
 *                by bucket range and runs rounds 2, 3 and the openings on slices when world is 4 or 8
 * mh_marlin_rccl_sliced(0) keeps the AHP rounds replicated (MSM sharding only); mh_marlin_set_shard / _set_alltoall with a
 * callback replace the native transport again; mh_marlin_rccl_destroy (also run by mh_shutdown) frees the communicator.
 * A rank that fails LOCALLY inside a sharded mh_marlin_prove (an allocation, a launch) fails the job on every rank: it keeps
 * entering the proof's collectives with a meaningless payload up to the next all-gather of partial points, whose error word makes
 * every rank return non-zero from the same commit round (the failing rank with its own message); the transport stays in step
 * and the next proof can run.  A rank that cannot enter a collective at all (a dead process, a sticky HIP error, the transport's
 * own staging buffers) still leaves its peers waiting: the caller must tear such a job down (torch.distributed.run does when a
 * rank exits non-zero). */
int mh_rccl_unique_id(uint8_t* id128_out);
int mh_marlin_set_rccl(int rank, int world, const uint8_t* id128);
int mh_marlin_rccl_sliced(int sliced);
int mh_marlin_rccl_destroy(void);
/* returns 1 when the native transport is active, 0 otherwise.  info4 (may be NULL): host-payload all-gathers, all-to-alls, device
 * all-gathers, bytes this rank sent to peers since mh_marlin_set_rccl; lib_path (may be NULL): the librccl the symbols came from */
int mh_marlin_rccl_info(uint64_t* info4, char* lib_path, size_t cap);
/* exchanges (any transport) since the last reset: how many, and the host's wall-clock milliseconds spent inside them */
int mh_marlin_exchange_stats(uint64_t* calls_out, double* host_ms_out, int reset);
int mh_ntt_dist_dev(int field, const void* d_in_local, void* d_out_local, uint32_t log_n, int inverse);
/* MSMs of C-layout slices: scalar i of job j multiplies base first_index[j] + i * stride of the handle's set (first_index =
 * rank + offset of the polynomial's base range, stride = world); needs the set's window table, which serves every rank's
 * slice as it is.  combine != 0: the partial points of all ranks are all-gathered and added (every rank returns the full
 * results, as after PC::commit); combine == 0: this rank's partial sums. */
int mh_msm_batch_sliced_dev(uint64_t bases_handle, size_t njobs, const size_t* first_index, size_t stride, const void* const* d_scalars_local,
                            const size_t* ns_local, int scalars_are_montgomery, int combine, uint64_t* out_xyz_mont);

/* Coefficients (Montgomery Fr) of a prover / indexer polynomial of the last proof made with this key, by the
 * reference's label ("w","z_a","z_b","mask_poly","t","g_1","h_1","g_2","h_2","row","col","a_val","b_val",
 * "c_val","row_col"; src/ahp/mod.rs:33-45).  out == NULL queries the length. */
int mh_marlin_get_poly(uint64_t pk, const char* label, uint64_t* out, size_t cap_elems, size_t* len_out);

/* ---- profiling: accumulated HIP-event time per kernel family on the library stream ----
 * family: 0 = ntt passes, 1 = msm (wall time of the MSM groups, all stages), 2 = msm accum only, 3 = glue, 4 = the msm sort and
 * bucket-reduction stages by themselves, 5 = the exchanges of a sharded proof (events on the library's stream around each
 * collective; with a host-synchronising callback transport the host's wait is in mh_marlin_exchange_stats instead), 6 = work the
 * prover runs on a second stream BESIDE an MSM batch's bucket reduction (the challenge-independent transforms of round 2 during
 * round 1's commitment): concurrent with family 1, so families 0 + 1 + 3 + 5 still add up to the step; 7 = the bucket reduction of the
 * fixed-base MSMs by itself (row / column sums and bit planes: part of family 4).
 * mh_prof_enable(on): 0 = off, 1 = every family, any other value = a mask with bit (f + 1) set for each family f to record
 * (8 = the accumulate kernel only).  An event pair per scope is not free: ~90 scopes per proof cost ~1 ms of launch gaps, so a
 * timed run records the one family it needs (bench.py) and takes the full breakdown from untimed proofs.  */
int mh_prof_enable(int on);
int mh_prof_reset(void);
int mh_prof_get(int family, double* total_ms_out, uint64_t* launches_out);

/* ---- invariants of the MSM pipeline (diagnostics; off by default) -----------------------------------------------------------
 * VariableBaseMSM::multi_scalar_mul (/root/reference src/lib.rs:172,193,213,292 via PC::commit / open_combinations) has exactly one
 * right answer.  With level >= 1 (or the environment variable MH_CHECK at mh_init) every fixed-base MSM batch checks each stage's
 * output against its input -- entries the scalars give = entries the split wrote = entries of the sorted lists, per partition;
 * bucket sizes and starts consistent; every bucket, row / column sum and plane on the curve (buffers pre-filled with garbage);
 * sum of rows = sum of columns = T plane; results on the curve -- and level 2 recomputes every list membership, every bucket (32-bit
 * complete law) and, for bucket sets up to 2^17, the whole reduction on the host.  A violation makes the MSM call (and a proof
 * that contains it) fail with MH_ECHECK naming the stage; mh_check_report returns the stage-by-stage text of the last checked
 * batch and, in counts2, the number of batches checked and of violations since mh_init. */
int mh_check_level(int level);
int mh_check_report(char* out, size_t cap, uint64_t* counts2);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* MARLIN_HIP_H */
