//! Seam B1, second scheme: `GpuSonicKZG10` -- `ark_poly_commit::sonic_pc::SonicKZG10` (what `benches/bench.rs:81`
//! instantiates; BASELINE configs[4] on BN254) with the MSMs of `commit` / `open` on the GPU.
//!
//! Restates ark-poly-commit 0.3 `sonic_pc/mod.rs` (SURVEY.md Appendix B-5 [UPSTREAM-RECALLED], confidence
//! medium-low): ONE `KZG10::commit` per polynomial -- against `shifted_powers(d)` (and the matching
//! `shifted_powers_of_gamma_g[d]`) when it has a degree bound d, else `powers` -- and one combined opening per
//! point on the unshifted powers; degree bounds are enforced with G2 elements at verification (upstream's
//! `check`, delegated).  Mirrors `marlin_amd/csrc/prover.hip` with `pc = 1` and `oracle/marlin.py: sonic_commit /
//! sonic_open`.
//!
//! UNCOMPILED (see Cargo.toml).
use crate::kzg::{kzg_commit, kzg_open_with_witness, srs_cache};
use crate::HipError;
use ark_bls12_381::{Bls12_381, Fr};
use ark_ff::Zero;
use ark_poly::univariate::DensePolynomial;
use ark_poly::{Polynomial, UVPolynomial};
use ark_poly_commit::sonic_pc::SonicKZG10;
use ark_poly_commit::{kzg10, Error as PCError, LabeledCommitment, LabeledPolynomial, PolynomialCommitment};
use ark_std::rand::RngCore;

type P = DensePolynomial<Fr>;
type Upstream = SonicKZG10<Bls12_381, P>;

/// `SonicKZG10<Bls12_381, DensePolynomial<Fr>>` whose multi-scalar multiplications run on the MI355X.
pub struct GpuSonicKZG10;

fn pc_err(e: HipError) -> PCError {
    PCError::IncorrectInputLength(e.to_string())
}

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

impl PolynomialCommitment<Fr, P> for GpuSonicKZG10 {
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

    /// `KZG10::setup(max_degree, produce_g2_powers = true)`: the only place G2 arithmetic appears on the setup path
    /// (`neg_powers_of_h`, one G2 scalar multiplication per enforceable bound) -- host, upstream.
    fn setup<R: RngCore>(max_degree: usize, num_vars: Option<usize>, rng: &mut R) -> Result<Self::UniversalParams, Self::Error> {
        Upstream::setup(max_degree, num_vars, rng)
    }

    fn trim(
        pp: &Self::UniversalParams,
        supported_degree: usize,
        supported_hiding_bound: usize,
        enforced_degree_bounds: Option<&[usize]>,
    ) -> Result<(Self::CommitterKey, Self::VerifierKey), Self::Error> {
        let (ck, vk) = Upstream::trim(pp, supported_degree, supported_hiding_bound, enforced_degree_bounds)?;
        srs_cache().get_or_upload(&ck.powers_of_g).map_err(pc_err)?;
        if let Some(sp) = ck.shifted_powers_of_g.as_ref() {
            srs_cache().get_or_upload(sp).map_err(pc_err)?;
        }
        Ok((ck, vk))
    }

    fn commit<'a>(
        ck: &Self::CommitterKey,
        polynomials: impl IntoIterator<Item = &'a LabeledPolynomial<Fr, P>>,
        rng: Option<&mut dyn RngCore>,
    ) -> Result<(Vec<LabeledCommitment<Self::Commitment>>, Vec<Self::Randomness>), Self::Error>
    where
        P: 'a,
    {
        let mut rng = rng;
        let srs = srs_cache().get_or_upload(&ck.powers_of_g).map_err(pc_err)?;
        let mut commitments = Vec::new();
        let mut randomness = Vec::new();
        for p in polynomials {
            let polynomial: &P = p.polynomial();
            let (comm, rand) = if let Some(d) = p.degree_bound() {
                // ck.shifted_powers(d): powers_of_g[max_degree - d ..] and shifted_powers_of_gamma_g[&d]
                let sp = ck.shifted_powers_of_g.as_ref().ok_or(PCError::UnsupportedDegreeBound(d))?;
                let ssrs = srs_cache().get_or_upload(sp).map_err(pc_err)?;
                let highest = *ck.enforced_degree_bounds.as_ref().and_then(|v| v.last()).ok_or(PCError::UnsupportedDegreeBound(d))?;
                let gamma = ck.shifted_powers_of_gamma_g.as_ref().and_then(|m| m.get(&d)).ok_or(PCError::UnsupportedDegreeBound(d))?;
                kzg_commit(&ssrs, sp, gamma, highest - d, polynomial, p.hiding_bound(), rng.as_mut().map(|r| &mut **r as &mut dyn RngCore))
            } else {
                kzg_commit(&srs, &ck.powers_of_g, &ck.powers_of_gamma_g, 0, polynomial, p.hiding_bound(), rng.as_mut().map(|r| &mut **r as &mut dyn RngCore))
            }
            .map_err(pc_err)?;
            commitments.push(LabeledCommitment::new(p.label().to_string(), comm, p.degree_bound()));
            randomness.push(rand);
        }
        Ok((commitments, randomness))
    }

    /// One combined polynomial (challenge `opening_challenges(i)` for the i-th polynomial), one `KZG10::open`.
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
        let srs = srs_cache().get_or_upload(&ck.powers_of_g).map_err(pc_err)?;
        let mut combined_polynomial = P::zero();
        let mut combined_rand = kzg10::Randomness::<Fr, P>::empty();
        let mut counter = 0u64;
        for (polynomial, rand) in labeled_polynomials.into_iter().zip(rands) {
            let ch = opening_challenges(counter);
            counter += 1;
            combined_polynomial += (ch, polynomial.polynomial());
            combined_rand += (ch, rand);
        }
        let witness = witness_polynomial(&combined_polynomial, *point);
        let hiding_witness = if combined_rand.is_hiding() {
            Some(witness_polynomial(&combined_rand.blinding_polynomial, *point))
        } else {
            None
        };
        kzg_open_with_witness(&srs, &ck.powers_of_g, &ck.powers_of_gamma_g, 0, *point, &combined_rand, &witness, hiding_witness.as_ref())
            .map_err(pc_err)
    }

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
    // batch_open / batch_check / open_combinations / check_combinations: the trait's defaults call the methods above
    // (SonicKZG10 overrides batch_check and check_combinations upstream only to batch pairings; the defaults are
    // equivalent and slower -- a maintainer may forward those two to `Upstream` exactly as GpuMarlinKZG10 does).
}
