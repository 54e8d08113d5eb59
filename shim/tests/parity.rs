//! The parity check this repository cannot run itself (no Rust toolchain in its image): the reference's own test
//! shapes (`/root/reference/src/test.rs:165-203`, `test_circuit`) proved on the stock CPU stack and on each GPU route
//! with identical inputs and identical rng streams; the `CanonicalSerialize` bytes of `Proof` and of
//! `IndexVerifierKey` must be equal.  A green run of this file is what turns "parity unpinned"  into
//! "pinned against arkworks".
//!
//!   MARLIN_HIP_LIB_DIR=../marlin_amd cargo test --release -- --test-threads=1
//!
//! `write_self_contained_golden_vectors` also writes `target/arkworks_golden.json`: per case the SRS (known tau and gamma
//! on the standard generators, so that a consumer regenerates it with `mh_srs_powers`; the small cases also embed the
//! compressed `powers_of_g` / `powers_of_gamma_g` for `mh_bases_upload_serialized`), the circuit parameters, every rng
//! seed, and the bytes arkworks produced (`to_bytes![vk]`, `proof.serialize(..)`).  Copy it to
//! `tests/golden/arkworks_golden.json`: `tests/test_arkworks_golden.py` then replays every case through the Python
//! oracle and through `mh_marlin_index` / `mh_marlin_prove` WITHOUT Rust and asserts equal bytes -- which is what pins
//! this repository's oracle to arkworks.
//!
//! UNCOMPILED (see Cargo.toml).
use ark_bls12_381::{Bls12_381, Fr};
use ark_ff::{Field, UniformRand};
use ark_marlin::{Marlin, SimpleHashFiatShamirRng};
use ark_poly::univariate::DensePolynomial;
use ark_poly_commit::marlin_pc::MarlinKZG10;
use ark_poly_commit::sonic_pc::SonicKZG10;
use ark_relations::lc;
use ark_relations::r1cs::{ConstraintSynthesizer, ConstraintSystemRef, SynthesisError};
use ark_serialize::CanonicalSerialize;
use ark_std::rand::SeedableRng;
use blake2::Blake2s;
use marlin_hip::marlin_pc::GpuMarlinKZG10;
use marlin_hip::prover::GpuMarlin;
use marlin_hip::sonic_pc::GpuSonicKZG10;
use rand_chacha::ChaChaRng;
use std::io::Write;

type FS = SimpleHashFiatShamirRng<Blake2s, ChaChaRng>;
type CpuMarlin = Marlin<Fr, MarlinKZG10<Bls12_381, DensePolynomial<Fr>>, FS>;
type GpuPcMarlin = Marlin<Fr, GpuMarlinKZG10, FS>;
type CpuSonic = Marlin<Fr, SonicKZG10<Bls12_381, DensePolynomial<Fr>>, FS>;
type GpuPcSonic = Marlin<Fr, GpuSonicKZG10, FS>;

/// `Circuit` of src/test.rs:10-52 (a * b = c repeated, then c * b = d; inputs c, d).
#[derive(Copy, Clone)]
struct Circuit<F: Field> {
    a: Option<F>,
    b: Option<F>,
    num_constraints: usize,
    num_variables: usize,
}

impl<F: Field> ConstraintSynthesizer<F> for Circuit<F> {
    fn generate_constraints(self, cs: ConstraintSystemRef<F>) -> Result<(), SynthesisError> {
        let a = cs.new_witness_variable(|| self.a.ok_or(SynthesisError::AssignmentMissing))?;
        let b = cs.new_witness_variable(|| self.b.ok_or(SynthesisError::AssignmentMissing))?;
        let c = cs.new_input_variable(|| Ok(self.a.unwrap() * self.b.unwrap()))?;
        let d = cs.new_input_variable(|| Ok(self.a.unwrap() * self.b.unwrap() * self.b.unwrap()))?;
        for _ in 0..(self.num_variables - 3) {
            let _ = cs.new_witness_variable(|| self.a.ok_or(SynthesisError::AssignmentMissing))?;
        }
        for _ in 0..(self.num_constraints - 1) {
            cs.enforce_constraint(lc!() + a, lc!() + b, lc!() + c)?;
        }
        cs.enforce_constraint(lc!() + c, lc!() + b, lc!() + d)?;
        Ok(())
    }
}

fn bytes<T: CanonicalSerialize>(t: &T) -> Vec<u8> {
    let mut v = Vec::new();
    t.serialize(&mut v).unwrap();
    v
}

fn hex(b: &[u8]) -> String {
    b.iter().map(|x| format!("{:02x}", x)).collect()
}

/// One shape: same SRS, same circuit, same zk seed on every stack.
fn run_shape(name: &str, num_constraints: usize, num_variables: usize, golden: &mut Vec<String>) {
    let mut setup_rng = ChaChaRng::from_seed([7u8; 32]);
    let srs = CpuMarlin::universal_setup(100, 25.max(num_variables), 300, &mut setup_rng).unwrap();
    let a = Fr::rand(&mut setup_rng);
    let b = Fr::rand(&mut setup_rng);
    let circ = Circuit { a: Some(a), b: Some(b), num_constraints, num_variables };
    let (c, d) = (a * b, a * b * b);
    let zk_seed = [42u8; 32];

    // stock stack
    let (cpu_pk, cpu_vk) = CpuMarlin::index(&srs, circ).unwrap();
    let cpu_proof = CpuMarlin::prove(&cpu_pk, circ, &mut ChaChaRng::from_seed(zk_seed)).unwrap();
    assert!(CpuMarlin::verify(&cpu_vk, &[c, d], &cpu_proof, &mut ChaChaRng::from_seed([1u8; 32])).unwrap());

    // seam B1 (+ B2 through the patched ark-poly): same Marlin code, GPU polynomial commitment
    let (gpu_pk, gpu_vk) = GpuPcMarlin::index(&srs, circ).unwrap();
    let gpu_proof = GpuPcMarlin::prove(&gpu_pk, circ, &mut ChaChaRng::from_seed(zk_seed)).unwrap();
    assert_eq!(bytes(&cpu_vk), bytes(&gpu_vk), "{}: vk bytes differ (seam B1)", name);
    assert_eq!(bytes(&cpu_proof), bytes(&gpu_proof), "{}: proof bytes differ (seam B1)", name);

    // whole-prover route: mh_marlin_index / mh_marlin_prove
    let (dev_pk, dev_vk) = GpuMarlin::index(&srs, circ).unwrap();
    let dev_proof = GpuMarlin::prove(&dev_pk, circ, zk_seed).unwrap();
    assert_eq!(bytes(&cpu_vk), bytes(&dev_vk), "{}: vk bytes differ (device prover)", name);
    assert_eq!(bytes(&cpu_proof), bytes(&dev_proof), "{}: proof bytes differ (device prover)", name);
    // and the stock verifier accepts / rejects it like its own (src/test.rs:158,161)
    assert!(CpuMarlin::verify(&cpu_vk, &[c, d], &dev_proof, &mut ChaChaRng::from_seed([1u8; 32])).unwrap());
    assert!(!CpuMarlin::verify(&cpu_vk, &[a, a], &dev_proof, &mut ChaChaRng::from_seed([1u8; 32])).unwrap());

    golden.push(format!(
        "{{\"name\":\"{}\",\"num_constraints\":{},\"num_variables\":{},\"a\":\"{}\",\"b\":\"{}\",\"vk\":\"{}\",\"proof\":\"{}\"}}",
        name, num_constraints, num_variables, hex(&bytes(&a)), hex(&bytes(&b)), hex(&bytes(&cpu_vk)), hex(&bytes(&cpu_proof))
    ));
}

#[test]
fn reference_test_shapes_are_byte_identical() {
    let mut golden = Vec::new();
    run_shape("tall_matrix_big", 100, 25, &mut golden);      // src/test.rs:166-171
    run_shape("tall_matrix_small", 26, 25, &mut golden);     // :174-179
    run_shape("squat_matrix_big", 25, 100, &mut golden);     // :182-187
    run_shape("squat_matrix_small", 25, 26, &mut golden);    // :190-195
    run_shape("square_matrix", 25, 25, &mut golden);         // :198-203
    let mut f = std::fs::File::create("target/golden_arkworks_shapes.json").unwrap();
    writeln!(f, "[{}]", golden.join(",\n")).unwrap();
}

// ---------------------------------------------------------------------------------------------------------------------
// Self-contained golden vectors for tests/test_arkworks_golden.py
// ---------------------------------------------------------------------------------------------------------------------

/// `KZG10::setup` for a KNOWN beta (= tau) and gamma on the standard generators -- what `kzg10::setup` computes after it
/// has drawn beta, g, gamma_g, h from its rng (ark-poly-commit 0.3 kzg10/mod.rs `setup`; reached from
/// `Marlin::universal_setup`, /root/reference src/lib.rs:79-96) -- so that the fixture is a few field elements instead of
/// megabytes of points, and a consumer without Rust rebuilds the same SRS with `mh_srs_powers`.  The proof and the key
/// are still produced by the stock `Marlin::{index, prove}` on this SRS.
fn known_tau_srs(max_degree: usize, tau: Fr, gamma: Fr, sonic: bool) -> ark_poly_commit::kzg10::UniversalParams<Bls12_381> {
    use ark_bls12_381::{G1Projective, G2Projective};
    use ark_ec::{msm::FixedBaseMSM, PairingEngine, ProjectiveCurve};
    use ark_ff::{One, PrimeField};
    use std::collections::BTreeMap;
    let g = G1Projective::prime_subgroup_generator();
    let gamma_g = g.mul(gamma.into_repr());
    let h = G2Projective::prime_subgroup_generator();
    let mut powers_of_tau = vec![Fr::one()];
    for i in 0..(max_degree + 1) {
        let next = powers_of_tau[i] * tau;
        powers_of_tau.push(next);
    }
    let scalar_bits = Fr::size_in_bits();
    let window = FixedBaseMSM::get_mul_window_size(max_degree + 2);
    let g_table = FixedBaseMSM::get_window_table(scalar_bits, window, g);
    let powers_of_g = FixedBaseMSM::multi_scalar_mul::<G1Projective>(scalar_bits, window, &g_table, &powers_of_tau[..=max_degree]);
    let gg_table = FixedBaseMSM::get_window_table(scalar_bits, window, gamma_g);
    let powers_of_gamma_g = FixedBaseMSM::multi_scalar_mul::<G1Projective>(scalar_bits, window, &gg_table, &powers_of_tau[..=max_degree + 1]);
    let powers_of_g = G1Projective::batch_normalization_into_affine(&powers_of_g);
    let powers_of_gamma_g: BTreeMap<usize, _> =
        G1Projective::batch_normalization_into_affine(&powers_of_gamma_g).into_iter().enumerate().collect();
    let mut neg_powers_of_h = BTreeMap::new();
    if sonic {
        // produce_g2_powers = true (SonicKZG10::setup): neg_powers_of_h[i] = [tau^-i] h, i <= max_degree
        let tau_inv = tau.inverse().unwrap();
        let mut cur = Fr::one();
        for i in 0..=max_degree {
            neg_powers_of_h.insert(i, h.mul(cur.into_repr()).into_affine());
            cur *= &tau_inv;
        }
    }
    let h_aff = h.into_affine();
    let beta_h = h.mul(tau.into_repr()).into_affine();
    ark_poly_commit::kzg10::UniversalParams {
        powers_of_g,
        powers_of_gamma_g,
        h: h_aff,
        beta_h,
        neg_powers_of_h,
        prepared_h: h_aff.into(),
        prepared_beta_h: beta_h.into(),
    }
}

/// `DummyCircuit` of benches/bench.rs:26-66.
#[derive(Copy, Clone)]
struct DummyCircuit<F: Field> {
    a: Option<F>,
    b: Option<F>,
    num_variables: usize,
    num_constraints: usize,
}

impl<F: Field> ConstraintSynthesizer<F> for DummyCircuit<F> {
    fn generate_constraints(self, cs: ConstraintSystemRef<F>) -> Result<(), SynthesisError> {
        let a = cs.new_witness_variable(|| self.a.ok_or(SynthesisError::AssignmentMissing))?;
        let b = cs.new_witness_variable(|| self.b.ok_or(SynthesisError::AssignmentMissing))?;
        let c = cs.new_input_variable(|| Ok(self.a.unwrap() * self.b.unwrap()))?;
        for _ in 0..(self.num_variables - 3) {
            let _ = cs.new_witness_variable(|| self.a.ok_or(SynthesisError::AssignmentMissing))?;
        }
        for _ in 0..self.num_constraints - 1 {
            cs.enforce_constraint(lc!() + a, lc!() + b, lc!() + c)?;
        }
        cs.enforce_constraint(lc!(), lc!(), lc!())?;
        Ok(())
    }
}

fn blake2s_hex(b: &[u8]) -> String {
    use digest::Digest;
    hex(&Blake2s::digest(b))
}

/// compressed points back to back, WITHOUT the Vec's u64 length prefix (what `mh_bases_upload_serialized` reads)
fn points_compressed<'a, I: Iterator<Item = &'a ark_bls12_381::G1Affine>>(pts: I) -> Vec<u8> {
    let mut v = Vec::new();
    for p in pts {
        p.serialize(&mut v).unwrap();
    }
    v
}

#[allow(clippy::too_many_arguments)]
fn golden_case<C: ConstraintSynthesizer<Fr> + Copy>(
    name: &str, sonic: bool, kind: &str, circ: C, public_input: &[Fr], nc: usize, nv: usize, a: Fr, b: Fr,
    setup: (usize, usize, usize), tau: Fr, gamma: Fr, zk_seed: [u8; 32], embed_srs: bool,
) -> String {
    use ark_ff::to_bytes;
    use ark_marlin::AHPForR1CS;
    let max_degree = AHPForR1CS::<Fr>::max_degree(setup.0, setup.1, setup.2).unwrap();   // src/lib.rs:86
    let srs = known_tau_srs(max_degree, tau, gamma, sonic);
    let (vk_bytes, proof_bytes) = if sonic {
        let (pk, vk) = CpuSonic::index(&srs, circ).unwrap();
        let proof = CpuSonic::prove(&pk, circ, &mut ChaChaRng::from_seed(zk_seed)).unwrap();
        assert!(CpuSonic::verify(&vk, public_input, &proof, &mut ChaChaRng::from_seed([1u8; 32])).unwrap());
        (to_bytes![vk].unwrap(), bytes(&proof))
    } else {
        let (pk, vk) = CpuMarlin::index(&srs, circ).unwrap();
        let proof = CpuMarlin::prove(&pk, circ, &mut ChaChaRng::from_seed(zk_seed)).unwrap();
        assert!(CpuMarlin::verify(&vk, public_input, &proof, &mut ChaChaRng::from_seed([1u8; 32])).unwrap());
        (to_bytes![vk].unwrap(), bytes(&proof))
    };
    let g_bytes = points_compressed(srs.powers_of_g.iter());
    let gg_bytes = points_compressed(srs.powers_of_gamma_g.values());
    let embedded = if embed_srs {
        format!(",\"powers_of_g\":\"{}\",\"powers_of_gamma_g\":\"{}\"", hex(&g_bytes), hex(&gg_bytes))
    } else {
        String::new()
    };
    format!(
        "{{\"name\":\"{}\",\"curve\":\"bls12_381\",\"pc\":\"{}\",\
          \"circuit\":{{\"kind\":\"{}\",\"num_constraints\":{},\"num_variables\":{},\"a\":\"{}\",\"b\":\"{}\"}},\
          \"srs\":{{\"num_constraints\":{},\"num_variables\":{},\"num_non_zero\":{},\"max_degree\":{},\"tau\":\"{}\",\"gamma\":\"{}\",\
          \"powers_of_g_blake2s\":\"{}\",\"powers_of_gamma_g_blake2s\":\"{}\"{}}},\
          \"zk_seed\":\"{}\",\"zk_rounds\":20,\"public_input\":[{}],\"vk_to_bytes\":\"{}\",\"proof\":\"{}\"}}",
        name, if sonic { "sonic" } else { "marlin" }, kind, nc, nv, hex(&bytes(&a)), hex(&bytes(&b)),
        setup.0, setup.1, setup.2, max_degree, hex(&bytes(&tau)), hex(&bytes(&gamma)), blake2s_hex(&g_bytes), blake2s_hex(&gg_bytes), embedded,
        hex(&zk_seed), public_input.iter().map(|x| format!("\"{}\"", hex(&bytes(x)))).collect::<Vec<_>>().join(","),
        hex(&vk_bytes), hex(&proof_bytes)
    )
}

/// The fixture `tests/test_arkworks_golden.py` consumes: the five `src/test.rs` shapes, DummyCircuit at 2^10 (BASELINE
/// configs[0]) on MarlinKZG10, and benches/bench.rs's own shape (2^16, SonicKZG10).
#[test]
fn write_self_contained_golden_vectors() {
    let mut rng = ChaChaRng::from_seed([11u8; 32]);
    let (tau, gamma) = (Fr::rand(&mut rng), Fr::rand(&mut rng));
    let (a, b) = (Fr::rand(&mut rng), Fr::rand(&mut rng));
    let zk_seed = [42u8; 32];
    let mut cases = Vec::new();
    for (name, nc, nv) in [("tall_matrix_big", 100usize, 25usize), ("tall_matrix_small", 26, 25), ("squat_matrix_big", 25, 100),
                           ("squat_matrix_small", 25, 26), ("square_matrix", 25, 25)] {
        let circ = Circuit { a: Some(a), b: Some(b), num_constraints: nc, num_variables: nv };
        for sonic in [false, true] {
            // src/test.rs:132: universal_setup(100, 25, 300); a squat matrix needs num_variables to cover its columns
            cases.push(golden_case(name, sonic, "test", circ, &[a * b, a * b * b], nc, nv, a, b, (100, nv.max(25), 300), tau, gamma, zk_seed, true));
        }
    }
    let n10 = 1usize << 10;
    let dummy10 = DummyCircuit { a: Some(a), b: Some(b), num_variables: 10, num_constraints: n10 };
    cases.push(golden_case("dummy_2p10", false, "dummy", dummy10, &[a * b], n10, 10, a, b, (n10, n10, 3 * n10), tau, gamma, zk_seed, true));
    let n16 = 1usize << 16;
    let dummy16 = DummyCircuit { a: Some(a), b: Some(b), num_variables: 10, num_constraints: n16 };
    // benches/bench.rs:75-83: universal_setup(65536, 65536, 3 * 65536), SonicKZG10; SRS regenerated from tau by the consumer
    cases.push(golden_case("bench_rs_dummy_2p16_sonic", true, "dummy", dummy16, &[a * b], n16, 10, a, b, (n16, n16, 3 * n16), tau, gamma, zk_seed, false));
    cases.push(golden_case("dummy_2p16", false, "dummy", dummy16, &[a * b], n16, 10, a, b, (n16, n16, 3 * n16), tau, gamma, zk_seed, false));
    let mut f = std::fs::File::create("target/arkworks_golden.json").unwrap();
    writeln!(f, "{{\"format\":1,\"producer\":\"arkworks 0.3 stock stack (shim/tests/parity.rs)\",\"cases\":[{}]}}", cases.join(",\n")).unwrap();
}

/// benches/bench.rs:81 instantiates SonicKZG10: the second PC scheme through seam B1.
#[test]
fn sonic_pc_is_byte_identical() {
    let mut rng = ChaChaRng::from_seed([9u8; 32]);
    let srs = CpuSonic::universal_setup(100, 25, 300, &mut rng).unwrap();
    let a = Fr::rand(&mut rng);
    let b = Fr::rand(&mut rng);
    let circ = Circuit { a: Some(a), b: Some(b), num_constraints: 100, num_variables: 25 };
    let (cpu_pk, cpu_vk) = CpuSonic::index(&srs, circ).unwrap();
    let (gpu_pk, gpu_vk) = GpuPcSonic::index(&srs, circ).unwrap();
    assert_eq!(bytes(&cpu_vk), bytes(&gpu_vk));
    let cpu_proof = CpuSonic::prove(&cpu_pk, circ, &mut ChaChaRng::from_seed([3u8; 32])).unwrap();
    let gpu_proof = GpuPcSonic::prove(&gpu_pk, circ, &mut ChaChaRng::from_seed([3u8; 32])).unwrap();
    assert_eq!(bytes(&cpu_proof), bytes(&gpu_proof));
}

/// `Marlin` is generic over `FS: FiatShamirRng` (src/lib.rs:64-70): the whole-prover route with the transcript supplied by the
/// caller (`mh_marlin_prove_fs`), once with the stock `SimpleHashFiatShamirRng<Blake2s, ChaChaRng>` routed through the callbacks
/// (must equal the built-in transcript AND the CPU) and once with a different digest (must equal the CPU prover instantiated
/// with that digest; the built-in transcript cannot produce it).
#[test]
fn caller_supplied_fiat_shamir_is_byte_identical() {
    type FsSha = SimpleHashFiatShamirRng<sha2::Sha256, ChaChaRng>;
    type CpuMarlinSha = Marlin<Fr, MarlinKZG10<Bls12_381, DensePolynomial<Fr>>, FsSha>;
    let mut rng = ChaChaRng::from_seed([11u8; 32]);
    let srs = CpuMarlin::universal_setup(100, 25, 300, &mut rng).unwrap();
    let a = Fr::rand(&mut rng);
    let b = Fr::rand(&mut rng);
    let circ = Circuit { a: Some(a), b: Some(b), num_constraints: 100, num_variables: 25 };
    let zk_seed = [42u8; 32];
    let (dev_pk, _dev_vk) = GpuMarlin::index(&srs, circ).unwrap();

    let (cpu_pk, cpu_vk) = CpuMarlin::index(&srs, circ).unwrap();
    let cpu_proof = CpuMarlin::prove(&cpu_pk, circ, &mut ChaChaRng::from_seed(zk_seed)).unwrap();
    let builtin = GpuMarlin::prove(&dev_pk, circ, zk_seed).unwrap();
    let routed = GpuMarlin::prove_with_fs::<_, FS>(&dev_pk, circ, zk_seed).unwrap();
    assert_eq!(bytes(&cpu_proof), bytes(&builtin));
    assert_eq!(bytes(&cpu_proof), bytes(&routed), "Blake2s transcript through the callbacks");

    // the index does not depend on FS (src/lib.rs:100-148 never touches fs_rng): the same device key serves both
    let (sha_pk, sha_vk) = CpuMarlinSha::index(&srs, circ).unwrap();
    assert_eq!(bytes(&cpu_vk), bytes(&sha_vk));
    let sha_cpu = CpuMarlinSha::prove(&sha_pk, circ, &mut ChaChaRng::from_seed(zk_seed)).unwrap();
    let sha_dev = GpuMarlin::prove_with_fs::<_, FsSha>(&dev_pk, circ, zk_seed).unwrap();
    assert_ne!(bytes(&sha_cpu), bytes(&cpu_proof));
    assert_eq!(bytes(&sha_cpu), bytes(&sha_dev), "Sha256 transcript through the callbacks");
    assert!(CpuMarlinSha::verify(&sha_vk, &[a * b, a * b * b], &sha_dev, &mut ChaChaRng::from_seed([1u8; 32])).unwrap());
}

/// Seam B2 alone: the patched radix-2 domain against the unpatched algorithm (which the patch keeps as
/// `in_order_fft_in_place_host`), forward / inverse / coset, 2^12 .. 2^20.
#[test]
fn patched_domain_matches_host() {
    use ark_poly::{EvaluationDomain, Radix2EvaluationDomain};
    let mut rng = ChaChaRng::from_seed([5u8; 32]);
    for log_n in [12u32, 13, 16, 20] {
        let d = Radix2EvaluationDomain::<Fr>::new(1 << log_n).unwrap();
        let v: Vec<Fr> = (0..(1usize << log_n)).map(|_| Fr::rand(&mut rng)).collect();
        let mut gpu = v.clone();
        d.fft_in_place(&mut gpu);
        let mut host = v.clone();
        d.fft_in_place_host(&mut host);
        assert_eq!(gpu, host, "fft 2^{}", log_n);
        d.ifft_in_place(&mut gpu);
        assert_eq!(gpu, v, "ifft round trip 2^{}", log_n);
        let mut cg = v.clone();
        d.coset_fft_in_place(&mut cg);
        let mut ch = v.clone();
        d.coset_fft_in_place_host(&mut ch);
        assert_eq!(cg, ch, "coset fft 2^{}", log_n);
        d.coset_ifft_in_place(&mut cg);
        assert_eq!(cg, v, "coset round trip 2^{}", log_n);
    }
}
