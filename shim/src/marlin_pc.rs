//! Seam B1: `GpuMarlinKZG10` -- `ark_poly_commit::marlin_pc::MarlinKZG10` with the MSMs of `commit` and `open`
//! on the GPU.  Every associated type IS upstream's, so keys, commitments and proofs made by either
//! implementation are interchangeable and `Marlin::verify` (which only calls `check_combinations`) is untouched.
//!
//! ```ignore
//! type MultiPC = GpuMarlinKZG10;                                     // src/test.rs:123 with one word changed
//! type MarlinInst = Marlin<Fr, MultiPC, SimpleHashFiatShamirRng<Blake2s, ChaChaRng>>;   // src/test.rs:128-130
//! ```
//!
//! Restates ark-poly-commit 0.3 `marlin_pc/mod.rs` (third-party, not in `/root/reference`; SURVEY.md Appendix
//! B-4 [UPSTREAM-RECALLED]) for `commit` and `open_individual_opening_challenges`; everything else delegates.
//! The same logic is what `marlin_amd/csrc/prover.hip: marlin_commit / open_at_point` and `oracle/marlin.py:
//! marlin_commit / marlin_open` implement, which is how it is tested in this repository.
//!
//! UNCOMPILED (see Cargo.toml).
use crate::kzg::{kzg_commit, kzg_open_with_witness, srs_cache};
use crate::HipError;
use ark_bls12_381::{Bls12_381, Fr, G1Affine};
use ark_ec::ProjectiveCurve;
use ark_ff::{One, Zero};
use ark_poly::univariate::DensePolynomial;
use ark_poly::{Polynomial, UVPolynomial};
use ark_poly_commit::marlin_pc::{self, MarlinKZG10};
use ark_poly_commit::{
    kzg10, BatchLCProof, Error as PCError, Evaluations, LabeledCommitment, LabeledPolynomial, LinearCombination,
    PCRandomness, PolynomialCommitment, QuerySet,
};
use ark_std::rand::RngCore;

type P = DensePolynomial<Fr>;
type Upstream = MarlinKZG10<Bls12_381, P>;

/// `MarlinKZG10<Bls12_381, DensePolynomial<Fr>>` whose multi-scalar multiplications run on the MI355X.
pub struct GpuMarlinKZG10;

fn pc_err(e: HipError) -> PCError {
    // upstream's catch-all for conditions its enum does not name
    PCError::IncorrectInputLength(e.to_string())
}

/// `p / (X - point)`, remainder dropped: what `KZG10::compute_witness_polynomial` keeps of the division
/// (`DenseOrSparsePolynomial::divide_with_q_and_r` by the linear divisor).
fn witness_polynomial(p: &P, point: Fr) -> P {
    let n = p.coeffs.len();
    if n <= 1 {
        return P::zero();
    }
    let mut q = vec![Fr::zero(); n - 1];
    let mut carry = Fr::zero();
    for i in (1..n).rev() {
        carry = p.coeffs[i] + carry * point;
        q[i - 1] = carry;
    }
    P::from_coefficients_vec(q)
}

impl PolynomialCommitment<Fr, P> for GpuMarlinKZG10 {
    type UniversalParams = <Upstream as PolynomialCommitment<Fr, P>>::UniversalParams;
    type CommitterKey = <Upstream as PolynomialCommitment<Fr, P>>::CommitterKey;
    type VerifierKey = <Upstream as PolynomialCommitment<Fr, P>>::VerifierKey;
    type PreparedVerifierKey = <Upstream as PolynomialCommitment<Fr, P>>::PreparedVerifierKey;
    type Commitment = <Upstream as PolynomialCommitment<Fr, P>>::Commitment;
    type PreparedCommitment = <Upstream as PolynomialCommitment<Fr, P>>::PreparedCommitment;
    type Randomness = <Upstream as PolynomialCommitment<Fr, P>>::Randomness;
    type Proof = <Upstream as PolynomialCommitment<Fr, P>>::Proof;
    type BatchProof = <Upstream as PolynomialCommitment<Fr, P>>::BatchProof;
    type Error = <Upstream as PolynomialCommitment<Fr, P>>::Error;

    /// `KZG10::setup` is unchanged (`Marlin::universal_setup`, `src/lib.rs:79-96`): the fixed-base powers are
    /// generated where upstream generates them; a known-tau bench SRS can be made on the device with
    /// `mh_srs_powers` instead (`prover::GpuMarlin::universal_setup_known_tau`).
    fn setup<R: RngCore>(max_degree: usize, num_vars: Option<usize>, rng: &mut R) -> Result<Self::UniversalParams, Self::Error> {
        Upstream::setup(max_degree, num_vars, rng)
    }

    /// `MarlinKZG10::trim`, then the device copies of `ck.powers` / `ck.shifted_powers` (+ the fixed-base window
    /// table) are made once, here, instead of inside the first `commit`.
    fn trim(
        pp: &Self::UniversalParams,
        supported_degree: usize,
        supported_hiding_bound: usize,
        enforced_degree_bounds: Option<&[usize]>,
    ) -> Result<(Self::CommitterKey, Self::VerifierKey), Self::Error> {
        let (ck, vk) = Upstream::trim(pp, supported_degree, supported_hiding_bound, enforced_degree_bounds)?;
        srs_cache().get_or_upload(&ck.powers).map_err(pc_err)?;
        if let Some(sp) = ck.shifted_powers.as_ref() {
            srs_cache().get_or_upload(sp).map_err(pc_err)?;
        }
        Ok((ck, vk))
    }

    /// `MarlinKZG10::commit` [B-4]: polynomials in order; per polynomial `KZG10::commit(powers)` and, when it has a
    /// degree bound d, a second `KZG10::commit(shifted_powers(d))` with fresh blinding draws.
    fn commit<'a>(
        ck: &Self::CommitterKey,
        polynomials: impl IntoIterator<Item = &'a LabeledPolynomial<Fr, P>>,
        rng: Option<&mut dyn RngCore>,
    ) -> Result<(Vec<LabeledCommitment<Self::Commitment>>, Vec<Self::Randomness>), Self::Error>
    where
        P: 'a,
    {
        let mut rng = rng;
        let srs = srs_cache().get_or_upload(&ck.powers).map_err(pc_err)?;
        let polynomials: Vec<&'a LabeledPolynomial<Fr, P>> = polynomials.into_iter().collect();
        // Pass 1 -- everything that touches the rng, in the reference's order (per polynomial: the blinding polynomial of the
        // commitment, then that of the shifted commitment).  The multi-scalar multiplications consume no randomness, so they
        // can run afterwards, all of them in ONE batched library call (kzg::msm_g1_batch -> mh_msm_batch).
        struct Plan {
            rand: kzg10::Randomness<Fr, P>,
            shifted: Option<(std::sync::Arc<crate::kzg::GpuSrs>, usize, kzg10::Randomness<Fr, P>)>,
        }
        let mut plans = Vec::with_capacity(polynomials.len());
        for p in &polynomials {
            let polynomial: &P = p.polynomial();
            // Error::check_degrees_and_bounds(ck.supported_degree(), ck.max_degree, enforced_degree_bounds, p)
            if polynomial.degree() > ck.powers.len() - 1 {
                return Err(PCError::TooManyCoefficients { num_coefficients: polynomial.degree() + 1, num_powers: ck.powers.len() });
            }
            let draw = |rng: &mut Option<&mut dyn RngCore>| -> Result<kzg10::Randomness<Fr, P>, PCError> {
                match p.hiding_bound() {
                    // Randomness::rand(hiding_bound, false, None, rng): blinding_polynomial = P::rand(hiding_bound + 1, rng)
                    Some(h) => Ok(kzg10::Randomness::rand(h, false, None, rng.as_mut().ok_or(PCError::MissingRng)?)),
                    None => Ok(kzg10::Randomness::empty()),
                }
            };
            let rand = draw(&mut rng)?;
            let shifted = if let Some(d) = p.degree_bound() {
                let sp = ck.shifted_powers.as_ref().ok_or(PCError::UnsupportedDegreeBound(d))?;
                let ssrs = srs_cache().get_or_upload(sp).map_err(pc_err)?;
                // ck.shifted_powers(d) = powers_of_g[max_degree - d ..]; sp starts at max_degree - highest bound
                let highest = *ck.enforced_degree_bounds.as_ref().and_then(|v| v.last()).ok_or(PCError::UnsupportedDegreeBound(d))?;
                Some((ssrs, highest - d, draw(&mut rng)?))
            } else {
                None
            };
            plans.push(Plan { rand, shifted });
        }
        // Pass 2 -- the large MSMs of the call as one batch (skip_leading_zeros as kzg10::commit does; short ones on the host)
        let mut jobs: Vec<(&crate::kzg::GpuSrs, usize, &[Fr])> = Vec::new();
        let mut slot: Vec<(Option<usize>, Option<usize>)> = Vec::new();
        for (p, plan) in polynomials.iter().zip(&plans) {
            let coeffs: &[Fr] = &p.polynomial().coeffs;
            let lz = coeffs.iter().take_while(|c| c.is_zero()).count();
            let tail = &coeffs[lz..];
            let big = tail.len() >= crate::kzg::GPU_MSM_THRESHOLD;
            let a = if big { jobs.push((&*srs, lz, tail)); Some(jobs.len() - 1) } else { None };
            let b = match (&plan.shifted, big) {
                (Some((ssrs, off, _)), true) => { jobs.push((&**ssrs, off + lz, tail)); Some(jobs.len() - 1) }
                _ => None,
            };
            slot.push((a, b));
        }
        let sums = crate::kzg::msm_g1_batch(&jobs).map_err(pc_err)?;
        // Pass 3 -- hiding parts (3 coefficients, host) and normalisation, per polynomial as upstream
        let mut commitments = Vec::new();
        let mut randomness = Vec::new();
        for ((p, plan), (a, b)) in polynomials.iter().zip(plans.into_iter()).zip(slot) {
            let coeffs: &[Fr] = &p.polynomial().coeffs;
            let finish = |sum: Option<usize>, host_powers: &[G1Affine], off: usize, rand: &kzg10::Randomness<Fr, P>| {
                let mut c = match sum {
                    Some(i) => sums[i],
                    None => crate::kzg::msm_host(&host_powers[off..off + coeffs.len()], coeffs),
                };
                let blind = &rand.blinding_polynomial.coeffs;
                if !blind.is_empty() {
                    c.add_assign_mixed(&crate::kzg::msm_host(&ck.powers_of_gamma_g[..blind.len()], blind).into_affine());
                }
                kzg10::Commitment(c.into_affine())
            };
            let comm = finish(a, &ck.powers, 0, &plan.rand);
            let (shifted_comm, shifted_rand) = match plan.shifted {
                Some((_, off, srand)) => {
                    let sp = ck.shifted_powers.as_ref().unwrap();
                    (Some(finish(b, sp, off, &srand)), Some(srand))
                }
                None => (None, None),
            };
            commitments.push(LabeledCommitment::new(p.label().to_string(), marlin_pc::Commitment { comm, shifted_comm }, p.degree_bound()));
            randomness.push(marlin_pc::Randomness { rand: plan.rand, shifted_rand });
        }
        Ok((commitments, randomness))
    }

    /// `MarlinKZG10::open_individual_opening_challenges` [B-4].  The challenge counter advances by one per
    /// polynomial and by one more per degree-bounded polynomial; the shifted witnesses are left-padded to the
    /// highest enforced bound and share ONE MSM over `shifted_powers(None)`; the two G1 results are added into a
    /// single `w`; `random_v` of the shifted proof is added only to a `Some` (a `None` from a non-hiding
    /// combination stays `None`).
    fn open_individual_opening_challenges<'a>(
        ck: &Self::CommitterKey,
        labeled_polynomials: impl IntoIterator<Item = &'a LabeledPolynomial<Fr, P>>,
        _commitments: impl IntoIterator<Item = &'a LabeledCommitment<Self::Commitment>>,
        point: &'a Fr,
        opening_challenges: &dyn Fn(u64) -> Fr,
        rands: impl IntoIterator<Item = &'a Self::Randomness>,
        _rng: Option<&mut dyn RngCore>,
    ) -> Result<Self::Proof, Self::Error>
    where
        P: 'a,
        Self::Randomness: 'a,
        Self::Commitment: 'a,
    {
        let point = *point;
        let srs = srs_cache().get_or_upload(&ck.powers).map_err(pc_err)?;
        let mut p = P::zero();
        let mut r = kzg10::Randomness::<Fr, P>::empty();
        let mut shifted_w = P::zero();
        let mut shifted_r = kzg10::Randomness::<Fr, P>::empty();
        let mut shifted_r_witness = P::zero();
        let mut enforce_degree_bound = false;
        let mut opening_challenge_counter = 0u64;
        for (polynomial, rand) in labeled_polynomials.into_iter().zip(rands) {
            let degree_bound = polynomial.degree_bound();
            let challenge_j = opening_challenges(opening_challenge_counter);
            opening_challenge_counter += 1;
            p += (challenge_j, polynomial.polynomial());
            r += (challenge_j, &rand.rand);
            if let Some(d) = degree_bound {
                enforce_degree_bound = true;
                let shifted_rand = rand.shifted_rand.as_ref().expect("degree-bounded polynomial without shifted randomness");
                let witness = witness_polynomial(polynomial.polynomial(), point);
                let shifted_rand_witness = if shifted_rand.is_hiding() {
                    Some(witness_polynomial(&shifted_rand.blinding_polynomial, point))
                } else {
                    None
                };
                let challenge_j_1 = opening_challenges(opening_challenge_counter);
                opening_challenge_counter += 1;
                let highest = *ck.enforced_degree_bounds.as_ref().and_then(|v| v.last()).ok_or(PCError::UnsupportedDegreeBound(d))?;
                // shift_polynomial(ck, &witness, d): left-pad with (highest bound - d) zero coefficients
                let shifted_witness = if witness.is_zero() {
                    P::zero()
                } else {
                    let mut c = vec![Fr::zero(); highest - d];
                    c.extend_from_slice(&witness.coeffs);
                    P::from_coefficients_vec(c)
                };
                shifted_w += (challenge_j_1, &shifted_witness);
                shifted_r += (challenge_j_1, shifted_rand);
                if let Some(srw) = shifted_rand_witness {
                    shifted_r_witness += (challenge_j_1, &srw);
                }
            }
        }
        // KZG10::open(&ck.powers(), &p, point, &r)
        let witness = witness_polynomial(&p, point);
        let hiding_witness = if r.is_hiding() { Some(witness_polynomial(&r.blinding_polynomial, point)) } else { None };
        let proof = kzg_open_with_witness(&srs, &ck.powers, &ck.powers_of_gamma_g, 0, point, &r, &witness, hiding_witness.as_ref())
            .map_err(pc_err)?;
        let mut w = proof.w.into_projective();
        let mut random_v = proof.random_v;
        if enforce_degree_bound {
            let sp = ck.shifted_powers.as_ref().expect("degree bounds enforced without shifted powers");
            let ssrs = srs_cache().get_or_upload(sp).map_err(pc_err)?;
            let shifted_proof = kzg_open_with_witness(
                &ssrs, sp, &ck.powers_of_gamma_g, 0, point, &shifted_r, &shifted_w, Some(&shifted_r_witness),
            ).map_err(pc_err)?;
            w += &shifted_proof.w.into_projective();
            if let Some(shifted_random_v) = shifted_proof.random_v {
                random_v = random_v.map(|v| v + &shifted_random_v);
            }
        }
        use ark_ec::{AffineCurve, ProjectiveCurve};
        Ok(kzg10::Proof { w: w.into_affine(), random_v })
    }

    // ---- verification: pairings only, no MSM of size n -- upstream, unchanged (src/lib.rs:413-423) ----
    fn check_individual_opening_challenges<'a>(
        vk: &Self::VerifierKey,
        commitments: impl IntoIterator<Item = &'a LabeledCommitment<Self::Commitment>>,
        point: &'a Fr,
        values: impl IntoIterator<Item = Fr>,
        proof: &Self::Proof,
        opening_challenges: &dyn Fn(u64) -> Fr,
        rng: Option<&mut dyn RngCore>,
    ) -> Result<bool, Self::Error>
    where
        Self::Commitment: 'a,
    {
        Upstream::check_individual_opening_challenges(vk, commitments, point, values, proof, opening_challenges, rng)
    }

    fn batch_check_individual_opening_challenges<'a, R: RngCore>(
        vk: &Self::VerifierKey,
        commitments: impl IntoIterator<Item = &'a LabeledCommitment<Self::Commitment>>,
        query_set: &QuerySet<Fr>,
        values: &Evaluations<Fr, Fr>,
        proof: &Self::BatchProof,
        opening_challenges: &dyn Fn(u64) -> Fr,
        rng: &mut R,
    ) -> Result<bool, Self::Error>
    where
        Self::Commitment: 'a,
    {
        Upstream::batch_check_individual_opening_challenges(vk, commitments, query_set, values, proof, opening_challenges, rng)
    }

    /// `MarlinKZG10::open_combinations_individual_opening_challenges` [B-4]: each LC becomes the dense polynomial
    /// sum(coeff * p) (`LCTerm::One` terms skipped), its randomness likewise, the degree bound kept only for
    /// single-term LCs, hiding bound = max; then the trait's `batch_open` groups the query set by point and calls
    /// `open` above.  The LC *commitments* upstream computes alongside are only consumed by the verifier's half
    /// and are not recomputed here.
    fn open_combinations_individual_opening_challenges<'a>(
        ck: &Self::CommitterKey,
        linear_combinations: impl IntoIterator<Item = &'a LinearCombination<Fr>>,
        polynomials: impl IntoIterator<Item = &'a LabeledPolynomial<Fr, P>>,
        commitments: impl IntoIterator<Item = &'a LabeledCommitment<Self::Commitment>>,
        query_set: &QuerySet<Fr>,
        opening_challenges: &dyn Fn(u64) -> Fr,
        rands: impl IntoIterator<Item = &'a Self::Randomness>,
        rng: Option<&mut dyn RngCore>,
    ) -> Result<BatchLCProof<Fr, P, Self>, Self::Error>
    where
        P: 'a,
        Self::Randomness: 'a,
        Self::Commitment: 'a,
    {
        use std::collections::BTreeMap;
        let label_map: BTreeMap<_, _> = polynomials
            .into_iter()
            .zip(rands)
            .zip(commitments)
            .map(|((p, r), c)| (p.label().clone(), (p, r, c)))
            .collect();
        let mut lc_polynomials = Vec::new();
        let mut lc_randomness = Vec::new();
        let mut lc_commitments = Vec::new();
        for lc in linear_combinations {
            let lc_label = lc.label().clone();
            let mut poly = P::zero();
            let mut degree_bound = None;
            let mut hiding_bound = None;
            let mut randomness = <Self::Randomness as PCRandomness>::empty();
            let mut comm = marlin_pc::Commitment::<Bls12_381>::default();
            let num_polys = lc.len();
            for (coeff, label) in lc.iter().filter(|(_, l)| !l.is_one()) {
                let label: &String = label.try_into().expect("cannot be one!");
                let &(cur_poly, cur_rand, cur_comm) =
                    label_map.get(label).ok_or(PCError::MissingPolynomial { label: label.to_string() })?;
                if num_polys == 1 && cur_poly.degree_bound().is_some() {
                    assert!(coeff.is_one(), "Coefficient must be one for degree-bounded equations");
                    degree_bound = cur_poly.degree_bound();
                } else if cur_poly.degree_bound().is_some() {
                    return Err(PCError::EquationHasDegreeBounds(lc_label));
                }
                hiding_bound = core::cmp::max(hiding_bound, cur_poly.hiding_bound());
                poly += (*coeff, cur_poly.polynomial());
                randomness += (*coeff, cur_rand);
                if num_polys == 1 {
                    comm = cur_comm.commitment().clone();   // single-term LC: the commitment itself
                }
            }
            lc_polynomials.push(LabeledPolynomial::new(lc_label.clone(), poly, degree_bound, hiding_bound));
            lc_randomness.push(randomness);
            lc_commitments.push(LabeledCommitment::new(lc_label, comm, degree_bound));
        }
        let proof = Self::batch_open_individual_opening_challenges(
            ck, lc_polynomials.iter(), lc_commitments.iter(), query_set, opening_challenges, lc_randomness.iter(), rng,
        )?;
        Ok(BatchLCProof { proof, evals: None })
    }

    fn check_combinations_individual_opening_challenges<'a, R: RngCore>(
        vk: &Self::VerifierKey,
        linear_combinations: impl IntoIterator<Item = &'a LinearCombination<Fr>>,
        commitments: impl IntoIterator<Item = &'a LabeledCommitment<Self::Commitment>>,
        query_set: &QuerySet<Fr>,
        evaluations: &Evaluations<Fr, Fr>,
        proof: &BatchLCProof<Fr, P, Self>,
        opening_challenges: &dyn Fn(u64) -> Fr,
        rng: &mut R,
    ) -> Result<bool, Self::Error>
    where
        Self::Commitment: 'a,
    {
        // same proof type, same vk: hand the pairing checks to upstream
        let upstream_proof = BatchLCProof::<Fr, P, Upstream> { proof: proof.proof.clone(), evals: proof.evals.clone() };
        Upstream::check_combinations_individual_opening_challenges(
            vk, linear_combinations, commitments, query_set, evaluations, &upstream_proof, opening_challenges, rng,
        )
    }
}
