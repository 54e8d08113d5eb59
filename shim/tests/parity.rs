//! The parity check this repository cannot run itself (no Rust toolchain in its image): the reference's own test
//! shapes (`/root/reference/src/test.rs:165-203`, `test_circuit`) proved on the stock CPU stack and on each GPU route
//! with identical inputs and identical rng streams; the `CanonicalSerialize` bytes of `Proof` and of
//! `IndexVerifierKey` must be equal.  A green run of this file is what turns "parity unpinned"  into
//! "pinned against arkworks".
//!
//!   MARLIN_HIP_LIB_DIR=../marlin_amd cargo test --release -- --test-threads=1
//!
//! It also writes `target/golden_arkworks.json` (hex of vk and proof per shape) so that the vectors can be committed
//! under `tests/golden/` of this repository and checked by `tests/test_gpu_marlin.py` without Rust.
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
    let mut f = std::fs::File::create("target/golden_arkworks.json").unwrap();
    writeln!(f, "[{}]", golden.join(",\n")).unwrap();
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
