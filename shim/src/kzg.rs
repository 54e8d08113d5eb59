//! `kzg10::KZG10::{commit, open}` with the multi-scalar multiplications on the GPU.
//!
//! Follows ark-poly-commit 0.3 `kzg10/mod.rs` (third-party; SURVEY.md Appendix B-3 [UPSTREAM-RECALLED]):
//! skip the low-order zero coefficients, `MSM(powers_of_g[lz..], coeffs)`, and when hiding draw
//! `P::rand(hiding_bound + 1, rng)` and add `MSM(powers_of_gamma_g, blinding)`.  The large MSM goes to
//! `mh_msm` against an SRS uploaded once (`GpuSrs`); the 3-coefficient hiding MSM stays on the host
//! (`VariableBaseMSM` upstream), exactly as this repository's `prover.hip: kzg_commit` does.
//!
//! UNCOMPILED (see Cargo.toml).
use crate::convert::{fr_slice_to_limbs, g1_from_jacobian_limbs, g1_slice_to_limbs};
use crate::{check, ensure_init, ffi, HipError};
use ark_bls12_381::{Bls12_381, Fr, G1Affine, G1Projective};
use ark_ec::msm::VariableBaseMSM;
use ark_ec::{AffineCurve, ProjectiveCurve};
use ark_ff::{PrimeField, Zero};
use ark_poly::univariate::DensePolynomial;
use ark_poly::UVPolynomial;
use ark_poly_commit::kzg10;
use ark_std::rand::RngCore;
use std::collections::HashMap;
use std::sync::Mutex;

/// MSMs shorter than this stay on the host: a launch + PCIe round trip costs more than ~2^10 host additions.
pub const GPU_MSM_THRESHOLD: usize = 1 << 10;

/// `powers_of_g` resident on the device.  ark-poly-commit slices ONE array (`powers[lz..]`,
/// `shifted_powers(d) = powers_of_g[max_degree - d ..]`), which is why `mh_msm` takes (handle, offset).
pub struct GpuSrs {
    pub handle: u64,
    pub len: usize,
}

impl GpuSrs {
    /// `mh_bases_upload` + `mh_bases_precompute`: after this every MSM against the handle runs the
    /// fixed-base path.  `PC::trim` is where upstream fixes the SRS slice, so that is where
    /// the wrappers call this.
    pub fn upload(powers_of_g: &[G1Affine], precompute: bool) -> Result<Self, HipError> {
        ensure_init();
        let limbs = g1_slice_to_limbs(powers_of_g);
        let mut handle = 0u64;
        check(unsafe { ffi::mh_bases_upload(ffi::MH_CURVE_BLS12_381_G1, limbs.as_ptr(), powers_of_g.len(), &mut handle) })?;
        if precompute && powers_of_g.len() >= (1 << 14) {
            check(unsafe { ffi::mh_bases_precompute(handle, 0) })?;
        }
        Ok(GpuSrs { handle, len: powers_of_g.len() })
    }
}

impl GpuSrs {
    /// `mh_bases_upload_serialized`: the SRS as it sits in a file -- `powers_of_g.serialize(..)` /
    /// `serialize_uncompressed(..)` of ark-serialize 0.3 WITHOUT the `Vec`'s u64 length prefix -- decoded (square root,
    /// sign flag) and validated on the device instead of by `Vec::<G1Affine>::deserialize` on the host.
    pub fn upload_serialized(bytes: &[u8], n: usize, compressed: bool, precompute: bool) -> Result<Self, HipError> {
        ensure_init();
        let item = if compressed { 48 } else { 96 };
        assert_eq!(bytes.len(), n * item, "serialized SRS: {} points need {} bytes", n, n * item);
        let mut handle = 0u64;
        check(unsafe { ffi::mh_bases_upload_serialized(ffi::MH_CURVE_BLS12_381_G1, bytes.as_ptr(), n, compressed as i32, &mut handle) })?;
        if precompute && n >= (1 << 14) {
            check(unsafe { ffi::mh_bases_precompute(handle, 0) })?;
        }
        Ok(GpuSrs { handle, len: n })
    }
}

impl Drop for GpuSrs {
    fn drop(&mut self) {
        unsafe { ffi::mh_bases_free(self.handle) };
    }
}

/// Process-wide cache: committer keys are plain upstream structs (no room for a handle), so the device copy of
/// `ck.powers` is looked up by the address and length of the slice.  Entries live as long as the process (an
/// `IndexProverKey` is proved against many times); `forget` drops one when its key is dropped.
pub(crate) struct SrsCache(Mutex<HashMap<(usize, usize), std::sync::Arc<GpuSrs>>>);

pub(crate) fn srs_cache() -> &'static SrsCache {
    static CACHE: std::sync::OnceLock<SrsCache> = std::sync::OnceLock::new();
    CACHE.get_or_init(|| SrsCache(Mutex::new(HashMap::new())))
}

impl SrsCache {
    pub(crate) fn get_or_upload(&self, powers: &[G1Affine]) -> Result<std::sync::Arc<GpuSrs>, HipError> {
        let key = (powers.as_ptr() as usize, powers.len());
        let mut m = self.0.lock().unwrap();
        if let Some(s) = m.get(&key) {
            return Ok(s.clone());
        }
        let s = std::sync::Arc::new(GpuSrs::upload(powers, true)?);
        m.insert(key, s.clone());
        Ok(s)
    }

    pub(crate) fn forget(&self, powers: &[G1Affine]) {
        self.0.lock().unwrap().remove(&(powers.as_ptr() as usize, powers.len()));
    }
}

/// Drop-in for `VariableBaseMSM::multi_scalar_mul(&powers_of_g[offset..offset + coeffs.len()], &repr(coeffs))`.
/// The scalars cross the boundary in Montgomery form (no `into_repr` pass on the host); the result is the same
/// group element arkworks computes (an MSM has one answer) in a different Jacobian representative, and every
/// commitment is normalised (`into_affine`) before it is hashed or serialised.
pub fn msm_g1(srs: &GpuSrs, offset: usize, coeffs: &[Fr]) -> Result<G1Projective, HipError> {
    assert!(offset + coeffs.len() <= srs.len, "MSM reads past the uploaded SRS");
    if coeffs.is_empty() {
        return Ok(G1Projective::zero());
    }
    let scalars = fr_slice_to_limbs(coeffs);
    let mut out = [0u64; 18];
    check(unsafe { ffi::mh_msm(srs.handle, offset, scalars.as_ptr(), 1, coeffs.len(), out.as_mut_ptr()) })?;
    Ok(g1_from_jacobian_limbs(&out))
}

/// Host MSM for the short ones (hiding terms: 3 coefficients on `powers_of_gamma_g`).
pub(crate) fn msm_host(bases: &[G1Affine], coeffs: &[Fr]) -> G1Projective {
    let repr: Vec<_> = coeffs.iter().map(|c| c.into_repr()).collect();
    VariableBaseMSM::multi_scalar_mul(bases, &repr)
}

fn skip_leading_zeros(coeffs: &[Fr]) -> usize {
    coeffs.iter().take_while(|c| c.is_zero()).count()
}

/// One unblinded MSM of a coefficient vector against `powers_of_g[offset..]` (kzg10 `commit` without the hiding
/// part): large ones on the device, short ones where upstream runs them.
pub(crate) fn commit_plain(srs: &GpuSrs, host_powers: &[G1Affine], offset: usize, coeffs: &[Fr]) -> Result<G1Projective, HipError> {
    let lz = skip_leading_zeros(coeffs);
    let tail = &coeffs[lz..];
    if tail.len() < GPU_MSM_THRESHOLD {
        Ok(msm_host(&host_powers[offset + lz..offset + coeffs.len()], tail))
    } else {
        msm_g1(srs, offset + lz, tail)
    }
}

/// `KZG10::commit(powers, polynomial, hiding_bound, rng)` [B-3].  `offset` selects `powers` (0) or
/// `shifted_powers(d)` (`max_degree - d`) inside the one uploaded array.
pub fn kzg_commit(
    srs: &GpuSrs,
    host_powers: &[G1Affine],
    powers_of_gamma_g: &[G1Affine],
    offset: usize,
    polynomial: &DensePolynomial<Fr>,
    hiding_bound: Option<usize>,
    rng: Option<&mut dyn RngCore>,
) -> Result<(kzg10::Commitment<Bls12_381>, kzg10::Randomness<Fr, DensePolynomial<Fr>>), HipError> {
    let mut commitment = commit_plain(srs, host_powers, offset, &polynomial.coeffs)?;
    let mut randomness = kzg10::Randomness::<Fr, DensePolynomial<Fr>>::empty();
    if let Some(hiding_degree) = hiding_bound {
        let rng = rng.ok_or(HipError::Unsupported("hiding commitment without an rng (kzg10::Error::MissingRng)"))?;
        // Randomness::rand(hiding_bound, false, None, rng): blinding_polynomial = P::rand(hiding_bound + 1, rng)
        randomness = kzg10::Randomness::rand(hiding_degree, false, None, rng);
        let blind = &randomness.blinding_polynomial.coeffs;
        let random_commitment = msm_host(&powers_of_gamma_g[..blind.len()], blind).into_affine();
        commitment.add_assign_mixed(&random_commitment);
    }
    Ok((kzg10::Commitment(commitment.into_affine()), randomness))
}

/// `KZG10::open_with_witness_polynomial` [B-4]: MSM of the witness, plus the blinding witness over
/// `powers_of_gamma_g` and `random_v = blinding(point)` when hiding.
pub fn kzg_open_with_witness(
    srs: &GpuSrs,
    host_powers: &[G1Affine],
    powers_of_gamma_g: &[G1Affine],
    offset: usize,
    point: Fr,
    randomness: &kzg10::Randomness<Fr, DensePolynomial<Fr>>,
    witness: &DensePolynomial<Fr>,
    hiding_witness: Option<&DensePolynomial<Fr>>,
) -> Result<kzg10::Proof<Bls12_381>, HipError> {
    use ark_poly::Polynomial;
    let mut w = commit_plain(srs, host_powers, offset, &witness.coeffs)?;
    let random_v = if let Some(hw) = hiding_witness {
        let blinding_evaluation = randomness.blinding_polynomial.evaluate(&point);
        w += &msm_host(&powers_of_gamma_g[..hw.coeffs.len()], &hw.coeffs);
        Some(blinding_evaluation)
    } else {
        None
    };
    Ok(kzg10::Proof { w: w.into_affine(), random_v })
}

/// Hook for the optional patched ark-ec (vendor/ark-ec-hip/msm_hip.patch): `Some(result)` when `G` is BLS12-381 G1
/// and the MSM is large enough for the device, `None` to let ark-ec's own Pippenger run.  `scalars` are canonical
/// `BigInteger256`s (what `VariableBaseMSM::multi_scalar_mul` receives), passed with `scalars_are_montgomery = 0`.
/// The base slice is resolved to (handle, offset) through `srs_cache`: slices of one `powers_of_g` allocation share
/// the upload made for the enclosing allocation's first sighting, later sub-slices are uploaded on demand.
pub fn msm_hook<G: AffineCurve + 'static>(
    bases: &[G],
    scalars: &[<G::ScalarField as PrimeField>::BigInt],
) -> Option<G::Projective> {
    use std::any::{Any, TypeId};
    let n = core::cmp::min(bases.len(), scalars.len());
    if TypeId::of::<G>() != TypeId::of::<G1Affine>() || n < GPU_MSM_THRESHOLD {
        return None;
    }
    // SAFETY: G == G1Affine was just checked
    let bases: &[G1Affine] = unsafe { core::slice::from_raw_parts(bases.as_ptr() as *const G1Affine, n) };
    let srs = srs_cache().get_or_upload(bases).ok()?;
    let mut limbs = Vec::with_capacity(n * 4);
    for s in &scalars[..n] {
        limbs.extend_from_slice(s.as_ref());
    }
    let mut out = [0u64; 18];
    check(unsafe { ffi::mh_msm(srs.handle, 0, limbs.as_ptr(), 0, n, out.as_mut_ptr()) }).ok()?;
    let p: G1Projective = g1_from_jacobian_limbs(&out);
    (Box::new(p) as Box<dyn Any>).downcast::<G::Projective>().ok().map(|b| *b)
}
