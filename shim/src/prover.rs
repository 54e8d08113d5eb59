//! The whole-prover route: `Marlin::{index, prove}` (`/root/reference/src/lib.rs:100-148, 151-311`) on the device
//! (`mh_marlin_index`, `mh_marlin_prove`), returning upstream's `IndexVerifierKey` / `Proof` so that the stock
//! `Marlin::verify` (`src/lib.rs:315-433`) checks the result.
//!
//! Host work that stays in Rust: constraint synthesis (ark-relations; `src/ahp/prover.rs:217-230`), which yields
//! the matrices (index) and the instance / witness assignments (prove), and the verifier key's `PC::trim`.
//!
//! UNCOMPILED (see Cargo.toml).
use crate::convert::{fr_slice_to_limbs, g1_slice_to_limbs};
use crate::{check, ensure_init, ffi, HipError};
use ark_bls12_381::{Bls12_381, Fr};
use ark_ff::{One, PrimeField, Zero};
use ark_marlin::ahp::indexer::Matrix;
use ark_marlin::{AHPForR1CS, IndexVerifierKey, Marlin, Proof, SimpleHashFiatShamirRng, UniversalSRS};
use ark_poly::univariate::DensePolynomial;
use ark_poly::{EvaluationDomain, GeneralEvaluationDomain};
use ark_poly_commit::marlin_pc::MarlinKZG10;
use ark_poly_commit::PolynomialCommitment;
use ark_relations::lc;
use ark_relations::r1cs::{ConstraintSynthesizer, ConstraintSystem, ConstraintSystemRef, OptimizationGoal, SynthesisMode};
use ark_serialize::CanonicalDeserialize;
use blake2::Blake2s;
use rand_chacha::ChaChaRng;

type P = DensePolynomial<Fr>;
pub type MultiPC = MarlinKZG10<Bls12_381, P>;
pub type FS = SimpleHashFiatShamirRng<Blake2s, ChaChaRng>;
pub type MarlinInst = Marlin<Fr, MultiPC, FS>;

/// Device-resident prover key (`IndexProverKey` upstream): SRS handles + the `mh_marlin_index` key.
pub struct GpuIndexProverKey {
    pk: u64,
    srs_g: u64,
    srs_gamma_g: u64,
    /// upstream verifier key for the same index, for `Marlin::verify`
    pub index_vk: IndexVerifierKey<Fr, MultiPC>,
}

impl Drop for GpuIndexProverKey {
    fn drop(&mut self) {
        unsafe {
            ffi::mh_marlin_pk_free(self.pk);
            ffi::mh_bases_free(self.srs_g);
            ffi::mh_bases_free(self.srs_gamma_g);
        }
    }
}

/// CSR image of one `Matrix<F> = Vec<Vec<(F, usize)>>` (`src/ahp/indexer.rs:81`).
struct Csr {
    row_ptr: Vec<u64>,
    col: Vec<u32>,
    val: Vec<u64>,
}

fn to_csr(m: &Matrix<Fr>) -> Csr {
    let mut row_ptr = Vec::with_capacity(m.len() + 1);
    let mut col = Vec::new();
    let mut vals = Vec::new();
    row_ptr.push(0u64);
    for row in m {
        for (v, j) in row {
            col.push(*j as u32);
            vals.push(*v);
        }
        row_ptr.push(col.len() as u64);
    }
    Csr { row_ptr, col, val: fr_slice_to_limbs(&vals) }
}

/// `pad_input_for_indexer_and_prover` + `make_matrices_square_for_prover`
/// (`src/ahp/constraint_systems.rs:45-81`; both `pub(crate)` upstream, restated here).
fn pad_and_square(cs: ConstraintSystemRef<Fr>) {
    let formatted_input_size = cs.num_instance_variables();
    let padded = GeneralEvaluationDomain::<Fr>::new(formatted_input_size).expect("domain_x").size();
    for _ in formatted_input_size..padded {
        cs.new_input_variable(|| Ok(Fr::zero())).unwrap();
    }
    cs.finalize();
    let num_variables = cs.num_instance_variables() + cs.num_witness_variables();
    let num_constraints = cs.num_constraints();
    if num_variables > num_constraints {
        for _ in 0..(num_variables - num_constraints) {
            cs.enforce_constraint(lc!(), lc!(), lc!()).expect("enforce 0 * 0 == 0 failed");
        }
    } else {
        for _ in 0..(num_constraints - num_variables) {
            let _ = cs.new_witness_variable(|| Ok(Fr::one())).expect("alloc failed");
        }
    }
}

pub struct GpuMarlin;

impl GpuMarlin {
    /// `Marlin::index` (`src/lib.rs:100-148`).  The AHP indexer's arithmetisation runs inside `mh_marlin_index`
    /// (6 iNTT + 6 MSM on the device); upstream's `AHPForR1CS::index` is called for the matrices and `index_info`
    /// only, and upstream's `PC::trim` for the (pairing-side) verifier key.
    pub fn index<C: ConstraintSynthesizer<Fr>>(
        srs: &UniversalSRS<Fr, MultiPC>,
        c: C,
    ) -> Result<(GpuIndexProverKey, IndexVerifierKey<Fr, MultiPC>), HipError> {
        ensure_init();
        let index = AHPForR1CS::<Fr>::index(c).map_err(|_| HipError::Unsupported("AHPForR1CS::index failed"))?;
        let info = index.index_info;
        let coeff_support = AHPForR1CS::<Fr>::get_degree_bounds(&info);
        let (_ck, verifier_key) = MultiPC::trim(srs, index.max_degree(), 1, Some(&coeff_support))
            .map_err(|_| HipError::Unsupported("PC::trim failed"))?;

        // whole powers_of_g (shifted powers index from the SRS's max_degree) + powers_of_gamma_g[0..3]
        let g = g1_slice_to_limbs(&srs.powers_of_g);
        let gamma: Vec<_> = (0..3).map(|i| srs.powers_of_gamma_g[&i]).collect();
        let gg = g1_slice_to_limbs(&gamma);
        let (mut srs_g, mut srs_gamma_g, mut pk) = (0u64, 0u64, 0u64);
        check(unsafe { ffi::mh_bases_upload(ffi::MH_CURVE_BLS12_381_G1, g.as_ptr(), srs.powers_of_g.len(), &mut srs_g) })?;
        check(unsafe { ffi::mh_bases_upload(ffi::MH_CURVE_BLS12_381_G1, gg.as_ptr(), 3, &mut srs_gamma_g) })?;

        let (a, b, cc) = (to_csr(&index.a), to_csr(&index.b), to_csr(&index.c));
        // instance variables come first in every row's column numbering (ark-relations `to_matrices`); after
        // pad_input_for_indexer_and_prover their count is already a power of two = |domain_x|
        let num_input = info.num_instance_variables;
        let m = ffi::mh_r1cs_matrices {
            num_constraints: info.num_constraints as u64,
            num_instance: num_input as u64,
            row_ptr: [a.row_ptr.as_ptr(), b.row_ptr.as_ptr(), cc.row_ptr.as_ptr()],
            col: [a.col.as_ptr(), b.col.as_ptr(), cc.col.as_ptr()],
            val: [a.val.as_ptr(), b.val.as_ptr(), cc.val.as_ptr()],
        };
        check(unsafe { ffi::mh_marlin_index(&m, srs_g, srs_gamma_g, &mut pk) })?;

        // IndexVerifierKey: index_info || 6 index commitments come back in ToBytes layout (mh_marlin_vk_bytes);
        // rebuild the commitments from them
        let mut len = 0usize;
        check(unsafe { ffi::mh_marlin_vk_bytes(pk, core::ptr::null_mut(), 0, &mut len) })?;
        let mut bytes = vec![0u8; len];
        check(unsafe { ffi::mh_marlin_vk_bytes(pk, bytes.as_mut_ptr(), len, &mut len) })?;
        let index_comms = crate::prover::wire::commitments_from_tobytes(&bytes[24..], 6);
        let index_vk = IndexVerifierKey { index_info: info, index_comms, verifier_key };
        Ok((GpuIndexProverKey { pk, srs_g, srs_gamma_g, index_vk: index_vk.clone() }, index_vk))
    }

    /// `Marlin::prove` (`src/lib.rs:151-311`).  `zk_rng` must be a ChaCha generator: its seed and round count
    /// cross the boundary so that the device reproduces the exact `Fp256::rand` stream.
    /// The caller passes the 32-byte seed it would have given `ChaChaRng::from_seed`.
    pub fn prove<C: ConstraintSynthesizer<Fr>>(
        pk: &GpuIndexProverKey,
        c: C,
        zk_seed: [u8; 32],
    ) -> Result<Proof<Fr, MultiPC>, HipError> {
        // witness synthesis, as AHPForR1CS::prover_init does (src/ahp/prover.rs:217-230)
        let pcs = ConstraintSystem::<Fr>::new_ref();
        pcs.set_optimization_goal(OptimizationGoal::Weight);
        pcs.set_mode(SynthesisMode::Prove { construct_matrices: true });
        c.generate_constraints(pcs.clone()).map_err(|_| HipError::Unsupported("constraint synthesis failed"))?;
        pad_and_square(pcs.clone());
        let pcs = pcs.into_inner().unwrap();
        let instance = fr_slice_to_limbs(&pcs.instance_assignment);
        let witness = fr_slice_to_limbs(&pcs.witness_assignment);

        let mut flat = vec![0u8; 4096];
        let mut flat_len = 0usize;
        check(unsafe {
            ffi::mh_marlin_prove(pk.pk, instance.as_ptr(), witness.as_ptr(), zk_seed.as_ptr(), 20, flat.as_mut_ptr(), flat.len(), &mut flat_len)
        })?;
        // flat ToBytes layout -> CanonicalSerialize bytes (host only) -> upstream's Proof
        let mut wire = vec![0u8; 4096];
        let mut wire_len = 0usize;
        check(unsafe { ffi::mh_marlin_proof_serialize(flat.as_ptr(), flat_len, 0, wire.as_mut_ptr(), wire.len(), &mut wire_len) })?;
        Proof::<Fr, MultiPC>::deserialize(&wire[..wire_len]).map_err(|_| HipError::Unsupported("Proof::deserialize rejected the library's bytes"))
    }

    /// `Marlin::<Fr, MultiPC, FS2>::prove` for ANY `FS2: FiatShamirRng` (`src/lib.rs:64-70,151-155`; the trait is
    /// `src/rng.rs:54-62`): the library's transcript operations are routed to `fs` through `mh_marlin_prove_fs` -- `initialize`
    /// (`src/lib.rs:161-163`), `absorb` after each round's commitments and after the evaluations (`:180,:201,:221,:289`) and
    /// `RngCore::next_u64` under `F::rand(fs_rng)` / `u128::rand(fs_rng)`.  `FS2::initialize` is a constructor upstream, so the
    /// state lives in an `Option<FS2>` that the first callback fills.
    pub fn prove_with_fs<C: ConstraintSynthesizer<Fr>, FS2: ark_marlin::rng::FiatShamirRng>(
        pk: &GpuIndexProverKey,
        c: C,
        zk_seed: [u8; 32],
    ) -> Result<Proof<Fr, MultiPC>, HipError> {
        use core::ffi::c_void;
        use ark_std::rand::RngCore;
        // The callbacks are `extern "C"`: a panic must not unwind through the library's C++ frames (undefined behaviour, an abort
        // at best).  Every callback runs under `catch_unwind`; the first failure -- a panic in the caller's `FS2`, or a callback
        // before `initialize` -- is recorded in the state, later callbacks become no-ops (`next_u64` returns 0), and
        // `prove_with_fs` returns `Err` once `mh_marlin_prove_fs` is back instead of a proof over a corrupted transcript.
        struct FsState<F> { fs: Option<F>, failed: bool }
        unsafe extern "C" fn init<F: ark_marlin::rng::FiatShamirRng>(user: *mut c_void, input: *const u8, len: usize) {
            let st = &mut *(user as *mut FsState<F>);
            if st.failed { return; }
            // Vec<u8> is what upstream's own call sites pass (`to_bytes![..]`, src/lib.rs:161)
            let bytes = core::slice::from_raw_parts(input, len).to_vec();
            match std::panic::catch_unwind(std::panic::AssertUnwindSafe(|| F::initialize(&bytes))) {
                Ok(fs) => st.fs = Some(fs),
                Err(_) => st.failed = true,
            }
        }
        unsafe extern "C" fn absorb<F: ark_marlin::rng::FiatShamirRng>(user: *mut c_void, input: *const u8, len: usize) {
            let st = &mut *(user as *mut FsState<F>);
            if st.failed { return; }
            let bytes = core::slice::from_raw_parts(input, len).to_vec();
            let ok = match st.fs.as_mut() {
                Some(fs) => std::panic::catch_unwind(std::panic::AssertUnwindSafe(|| fs.absorb(&bytes))).is_ok(),
                None => false,
            };
            if !ok { st.failed = true; }
        }
        unsafe extern "C" fn next<F: ark_marlin::rng::FiatShamirRng>(user: *mut c_void) -> u64 {
            let st = &mut *(user as *mut FsState<F>);
            if st.failed { return 0; }
            let r = match st.fs.as_mut() {
                Some(fs) => std::panic::catch_unwind(std::panic::AssertUnwindSafe(|| fs.next_u64())).ok(),
                None => None,
            };
            match r { Some(v) => v, None => { st.failed = true; 0 } }
        }
        let pcs = ConstraintSystem::<Fr>::new_ref();
        pcs.set_optimization_goal(OptimizationGoal::Weight);
        pcs.set_mode(SynthesisMode::Prove { construct_matrices: true });
        c.generate_constraints(pcs.clone()).map_err(|_| HipError::Unsupported("constraint synthesis failed"))?;
        pad_and_square(pcs.clone());
        let pcs = pcs.into_inner().unwrap();
        let instance = fr_slice_to_limbs(&pcs.instance_assignment);
        let witness = fr_slice_to_limbs(&pcs.witness_assignment);
        let mut state: FsState<FS2> = FsState { fs: None, failed: false };
        let cb = ffi::mh_fiat_shamir {
            user: &mut state as *mut FsState<FS2> as *mut c_void,
            initialize: Some(init::<FS2>),
            absorb: Some(absorb::<FS2>),
            next_u64: Some(next::<FS2>),
        };
        let mut flat = vec![0u8; 4096];
        let mut flat_len = 0usize;
        let rc = unsafe {
            ffi::mh_marlin_prove_fs(pk.pk, instance.as_ptr(), witness.as_ptr(), zk_seed.as_ptr(), 20, &cb, flat.as_mut_ptr(), flat.len(), &mut flat_len)
        };
        if state.failed {
            return Err(HipError::Unsupported("the caller's FiatShamirRng panicked (or was used before initialize) inside a transcript callback"));
        }
        check(rc)?;
        let mut wire = vec![0u8; 4096];
        let mut wire_len = 0usize;
        check(unsafe { ffi::mh_marlin_proof_serialize(flat.as_ptr(), flat_len, 0, wire.as_mut_ptr(), wire.len(), &mut wire_len) })?;
        Proof::<Fr, MultiPC>::deserialize(&wire[..wire_len]).map_err(|_| HipError::Unsupported("Proof::deserialize rejected the library's bytes"))
    }
}

// ---- N GPUs behind ONE `Marlin::prove` -------------------------------------------------------------
//
// The reference has one caller (`src/lib.rs:151-155`) and rayon threads under it (`src/ahp/mod.rs:9-10`).  Two ways to keep that
// shape with N GPUs:
//   * `GpuGroup` -- ONE process: a context per GPU joined by the library's in-process transport (`mh_group_create`); `index` and
//     `prove` run rank r's share on a library thread bound to context r (`mh_group_run`) and hand back rank 0's proof after
//     checking that every rank produced the same bytes.  No RCCL, no launcher.
//   * `GpuMarlin::join_rccl(rank, world, id)` -- one PROCESS per GPU (an MPI-style launcher): after it, the ordinary
//     `GpuMarlin::{index, prove}` of that process are rank `rank` of a sharded prover over the library's own RCCL communicator.
// Every rank must be given identical arguments (the same circuit, the same `zk_seed`); a rank that fails makes `prove` return
// `Err` on EVERY rank from the same commit round (the error word travels with the partial points).

/// One process, N GPUs.
pub struct GpuGroup {
    group: ffi::mh_group_t,
    world: usize,
}
unsafe impl Send for GpuGroup {}
unsafe impl Sync for GpuGroup {}

/// Per-rank prover keys of a group (each lives in its rank's context and is freed there).
pub struct GpuGroupKey<'g> {
    group: &'g GpuGroup,
    keys: Vec<Option<GpuIndexProverKey>>,
    pub index_vk: IndexVerifierKey<Fr, MultiPC>,
}

impl GpuGroup {
    /// `devices[r]` = the HIP device of rank r (a device may repeat).
    pub fn new(devices: &[i32]) -> Result<Self, HipError> {
        let mut group: ffi::mh_group_t = core::ptr::null_mut();
        check(unsafe { ffi::mh_group_create(devices.as_ptr(), devices.len() as i32, &mut group) })?;
        Ok(GpuGroup { group, world: devices.len() })
    }
    pub fn world(&self) -> usize { self.world }

    /// `f(rank)` on `world` library threads, thread r bound to context r; the first failing rank's error comes back.
    fn run<T: Send, F: Fn(usize) -> Result<T, HipError> + Sync>(&self, f: F) -> Result<Vec<T>, HipError> {
        use core::ffi::c_void;
        struct Job<'a, T, F> { f: &'a F, out: Vec<std::sync::Mutex<Option<Result<T, HipError>>>> }
        unsafe extern "C" fn tramp<T: Send, F: Fn(usize) -> Result<T, HipError> + Sync>(rank: i32, user: *mut c_void) -> i32 {
            let job = &*(user as *const Job<T, F>);
            // a panic must not unwind into the library's C++ frames
            let r = std::panic::catch_unwind(std::panic::AssertUnwindSafe(|| (job.f)(rank as usize)))
                .unwrap_or(Err(HipError::Unsupported("a rank panicked")));
            let rc = if r.is_ok() { 0 } else { ffi::MH_EHIP };
            *job.out[rank as usize].lock().unwrap() = Some(r);
            rc
        }
        let job = Job { f: &f, out: (0..self.world).map(|_| std::sync::Mutex::new(None)).collect() };
        let cb: ffi::mh_group_fn = Some(tramp::<T, F>);
        let user = &job as *const Job<T, F> as *mut c_void;
        let rc = unsafe { ffi::mh_group_run(self.group, cb, user) };
        let mut res = Vec::with_capacity(self.world);
        for slot in job.out {
            match slot.into_inner().unwrap() {
                Some(Ok(v)) => res.push(v),
                Some(Err(e)) => return Err(e),
                None => { check(rc)?; return Err(HipError::Unsupported("a rank did not run")); }
            }
        }
        Ok(res)
    }

    /// `Marlin::index` on every rank (the index is never sharded: every GPU holds the whole key and the whole window table).
    pub fn index<C: ConstraintSynthesizer<Fr> + Clone + Sync>(
        &self,
        srs: &UniversalSRS<Fr, MultiPC>,
        c: C,
    ) -> Result<GpuGroupKey<'_>, HipError> {
        let keys = self.run(|_rank| GpuMarlin::index(srs, c.clone()).map(|(pk, _vk)| pk))?;
        let index_vk = keys[0].index_vk.clone();
        Ok(GpuGroupKey { group: self, keys: keys.into_iter().map(Some).collect(), index_vk })
    }

    /// ONE `Marlin::prove` on all GPUs of the group: bucket-range-sharded MSMs, sliced rounds from 4 ranks on.
    pub fn prove<C: ConstraintSynthesizer<Fr> + Clone + Sync>(
        &self,
        key: &GpuGroupKey<'_>,
        c: C,
        zk_seed: [u8; 32],
    ) -> Result<Proof<Fr, MultiPC>, HipError> {
        use ark_serialize::CanonicalSerialize;
        let proofs = self.run(|rank| GpuMarlin::prove(key.keys[rank].as_ref().unwrap(), c.clone(), zk_seed))?;
        let bytes = |p: &Proof<Fr, MultiPC>| { let mut v = Vec::new(); p.serialize(&mut v).unwrap(); v };
        let first = bytes(&proofs[0]);
        if proofs.iter().skip(1).any(|p| bytes(p) != first) {
            return Err(HipError::Unsupported("the ranks of a sharded proof disagree"));
        }
        Ok(proofs.into_iter().next().unwrap())
    }
}

impl Drop for GpuGroupKey<'_> {
    fn drop(&mut self) {
        // each key is freed inside its own context (handles belong to the context that made them)
        let keys: Vec<std::sync::Mutex<Option<GpuIndexProverKey>>> = self.keys.drain(..).map(std::sync::Mutex::new).collect();
        let _ = self.group.run(|rank| { drop(keys[rank].lock().unwrap().take()); Ok(()) });
    }
}

impl Drop for GpuGroup {
    fn drop(&mut self) { unsafe { ffi::mh_group_destroy(self.group); } }
}

impl GpuMarlin {
    /// Rank 0 of a process-per-GPU job draws the communicator's id; the caller's launcher hands the 128 bytes to every rank.
    pub fn rccl_unique_id() -> Result<[u8; 128], HipError> {
        let mut id = [0u8; 128];
        check(unsafe { ffi::mh_rccl_unique_id(id.as_mut_ptr()) })?;
        Ok(id)
    }

    /// Collective over the `world` processes of the job: afterwards this process's `GpuMarlin::{index, prove}` are rank `rank` of
    /// a sharded prover (`mh_marlin_set_rccl`).  Call after `mh_init(device)` (`ensure_init`) and before `index`.
    pub fn join_rccl(rank: usize, world: usize, id: &[u8; 128]) -> Result<(), HipError> {
        ensure_init();
        check(unsafe { ffi::mh_marlin_set_rccl(rank as i32, world as i32, id.as_ptr()) })
    }

    /// `Marlin::prove` as rank `rank` of `world` processes, in one call: joins the communicator if this process has not yet, then
    /// proves.  Every rank returns the same proof (or every rank returns `Err`).
    pub fn prove_sharded<C: ConstraintSynthesizer<Fr>>(
        pk: &GpuIndexProverKey,
        c: C,
        zk_seed: [u8; 32],
        rank: usize,
        world: usize,
        id: &[u8; 128],
    ) -> Result<Proof<Fr, MultiPC>, HipError> {
        let mut info = [0u64; 4];
        let active = unsafe { ffi::mh_marlin_rccl_info(info.as_mut_ptr(), core::ptr::null_mut(), 0) };
        if active == 0 { Self::join_rccl(rank, world, id)?; }
        Self::prove(pk, c, zk_seed)
    }
}

/// `ToBytes` images (uncompressed, with presence bytes) -> upstream commitment structs.
pub mod wire {
    use ark_bls12_381::{Bls12_381, Fq, G1Affine};
    use ark_ff::{BigInteger384, FromBytes, PrimeField};
    use ark_poly_commit::{kzg10, marlin_pc};

    fn g1_from_tobytes(b: &[u8]) -> G1Affine {
        // x (48 B LE canonical) || y (48 B) || infinity (1 B)   [SURVEY.md Appendix B-6]
        let x = Fq::from_repr(BigInteger384::read(&b[0..48]).unwrap()).unwrap();
        let y = Fq::from_repr(BigInteger384::read(&b[48..96]).unwrap()).unwrap();
        G1Affine::new(x, y, b[96] != 0)
    }

    /// `marlin_pc::Commitment::write`: comm (97 B) || shifted_comm.is_some() (1 B) || shifted or identity (97 B).
    pub fn commitments_from_tobytes(bytes: &[u8], n: usize) -> Vec<marlin_pc::Commitment<Bls12_381>> {
        (0..n)
            .map(|i| {
                let b = &bytes[195 * i..195 * (i + 1)];
                let comm = kzg10::Commitment(g1_from_tobytes(&b[0..97]));
                let shifted_comm = if b[97] != 0 { Some(kzg10::Commitment(g1_from_tobytes(&b[98..195]))) } else { None };
                marlin_pc::Commitment { comm, shifted_comm }
            })
            .collect()
    }
}
