//! `kzg10::KZG10::{commit, open}` with the multi-scalar multiplications on the GPU.
//!
//! Follows ark-poly-commit 0.3 `kzg10/mod.rs` (third-party; SURVEY.md Appendix B-3 [UPSTREAM-RECALLED]):
//! skip the low-order zero coefficients, `MSM(powers_of_g[lz..], coeffs)`, and when hiding draw
//! `P::rand(hiding_bound + 1, rng)` and add `MSM(powers_of_gamma_g, blinding)`.  The large MSM goes to
//! `mh_msm` against an SRS uploaded once (`GpuSrs`); the 3-coefficient hiding MSM stays on the host
//! (`VariableBaseMSM` upstream), exactly as this repository's `prover.hip: kzg_commit` does.
//!
//! UNCOMPILED (see Cargo.toml).
use crate::convert::{fr_slice_to_limbs, g1_from_jacobian_limbs};
use crate::{check, ensure_init, ffi, HipError};
use ark_bls12_381::{Bls12_381, Fr, G1Affine, G1Projective};
use ark_ec::msm::VariableBaseMSM;
use ark_ec::ProjectiveCurve;
use ark_ff::{PrimeField, Zero};
use ark_poly::univariate::DensePolynomial;
use ark_poly::UVPolynomial;
use ark_poly_commit::kzg10;
use ark_std::rand::RngCore;
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
        // an SRS never contains the identity; a base set that does cannot cross this boundary (no infinity flag)
        let limbs = crate::convert::g1_slice_to_limbs_checked(powers_of_g)
            .ok_or(HipError::Unsupported("base set contains the identity"))?;
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

/// Process-wide cache of device copies of `ck.powers`.  Committer keys are plain upstream structs (no room for a handle:
/// `Marlin`'s `IndexProverKey` embeds `PC::CommitterKey` as is), so the device copy is found through the slice itself.
/// The key is (address, length) PLUS a fingerprint of the contents -- the points at indices 0, 1, len / 2, len - 1 --
/// checked on every lookup: a `Vec` that was freed and whose address and length were reused by a different base vector
/// (another tau: every point differs) misses, is uploaded afresh and replaces the stale entry.  The cache is bounded
/// (`MAX_ENTRIES`, least recently used out first; the device memory goes when the last `Arc` does), so distinct keys do
/// not accumulate uploads and window tables for the life of the process.
pub(crate) struct SrsCache(Mutex<Vec<CacheEntry>>);
struct CacheEntry {
    ptr: usize,
    len: usize,
    fingerprint: [G1Affine; 4],
    srs: std::sync::Arc<GpuSrs>,
    last_use: u64,
}
const MAX_ENTRIES: usize = 8;

fn fingerprint(powers: &[G1Affine]) -> [G1Affine; 4] {
    let n = powers.len();
    [powers[0], powers[1.min(n - 1)], powers[n / 2], powers[n - 1]]
}

pub(crate) fn srs_cache() -> &'static SrsCache {
    static CACHE: std::sync::OnceLock<SrsCache> = std::sync::OnceLock::new();
    CACHE.get_or_init(|| SrsCache(Mutex::new(Vec::new())))
}

static CLOCK: std::sync::atomic::AtomicU64 = std::sync::atomic::AtomicU64::new(0);

impl SrsCache {
    /// The device copy of `powers` (uploading and building the window table on a miss).  Called by the wrappers'
    /// `trim` -- where upstream fixes the SRS slice -- and again by `commit` / `open`, which then hit.
    pub(crate) fn get_or_upload(&self, powers: &[G1Affine]) -> Result<std::sync::Arc<GpuSrs>, HipError> {
        assert!(!powers.is_empty(), "empty committer key");
        let (ptr, len, fp) = (powers.as_ptr() as usize, powers.len(), fingerprint(powers));
        let now = CLOCK.fetch_add(1, std::sync::atomic::Ordering::Relaxed);
        let mut m = self.0.lock().unwrap();
        if let Some(i) = m.iter().position(|e| e.ptr == ptr && e.len == len) {
            if m[i].fingerprint == fp {
                m[i].last_use = now;
                return Ok(m[i].srs.clone());
            }
            m.swap_remove(i); // same address and length, other contents: the old vector is gone
        }
        let srs = std::sync::Arc::new(GpuSrs::upload(powers, true)?);
        if m.len() >= MAX_ENTRIES {
            let oldest = (0..m.len()).min_by_key(|&i| m[i].last_use).unwrap();
            m.swap_remove(oldest);
        }
        m.push(CacheEntry { ptr, len, fingerprint: fp, srs: srs.clone(), last_use: now });
        Ok(srs)
    }

    /// The entry whose allocation CONTAINS `slice`, as (device copy, offset) -- never uploads.  Used by the transparent
    /// MSM hook, which must not create device state for arbitrary base slices.
    pub(crate) fn resolve_containing(&self, slice: &[G1Affine]) -> Option<(std::sync::Arc<GpuSrs>, usize)> {
        let (lo, bytes) = (slice.as_ptr() as usize, core::mem::size_of::<G1Affine>());
        let m = self.0.lock().unwrap();
        for e in m.iter() {
            if lo >= e.ptr && (lo - e.ptr) % bytes == 0 {
                let off = (lo - e.ptr) / bytes;
                if off + slice.len() <= e.len {
                    return Some((e.srs.clone(), off));
                }
            }
        }
        None
    }

    pub(crate) fn forget(&self, powers: &[G1Affine]) {
        let (ptr, len) = (powers.as_ptr() as usize, powers.len());
        self.0.lock().unwrap().retain(|e| !(e.ptr == ptr && e.len == len));
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

/// Several MSMs in ONE library call (`mh_msm_batch`): the polynomials of one `PC::commit` go up back to back and share one
/// sort / accumulate / reduce sequence on the device; a coefficient slice that appears twice (a degree-bounded polynomial
/// against `powers` and against `shifted_powers`) is uploaded once when both jobs name the same SRS handle.  Jobs are
/// `(srs, offset, coefficients)`; results in job order.
pub fn msm_g1_batch(jobs: &[(&GpuSrs, usize, &[Fr])]) -> Result<Vec<G1Projective>, HipError> {
    if jobs.is_empty() {
        return Ok(Vec::new());
    }
    // the library reads Montgomery limbs; `Fr` is `repr(Rust)`, so each vector is marshalled once (convert.rs)
    let limbs: Vec<Vec<u64>> = jobs.iter().map(|(_, _, c)| fr_slice_to_limbs(c)).collect();
    let handles: Vec<u64> = jobs.iter().map(|(s, _, _)| s.handle).collect();
    let offsets: Vec<usize> = jobs.iter().map(|(_, o, _)| *o).collect();
    let ns: Vec<usize> = jobs.iter().map(|(_, _, c)| c.len()).collect();
    // same host slice (pointer and length) -> same marshalled buffer, so that the library sees the repeated vector
    let mut ptrs: Vec<*const u64> = Vec::with_capacity(jobs.len());
    for (j, (_, _, c)) in jobs.iter().enumerate() {
        let first = jobs[..j].iter().position(|(_, _, d)| d.as_ptr() == c.as_ptr() && d.len() == c.len());
        ptrs.push(match first {
            Some(k) => limbs[k].as_ptr(),
            None => limbs[j].as_ptr(),
        });
    }
    for (j, (s, o, c)) in jobs.iter().enumerate() {
        assert!(o + c.len() <= s.len, "MSM {} reads past the uploaded SRS", j);
    }
    let mut out = vec![0u64; 18 * jobs.len()];
    check(unsafe { ffi::mh_msm_batch(jobs.len(), handles.as_ptr(), offsets.as_ptr(), ptrs.as_ptr(), ns.as_ptr(), 1, out.as_mut_ptr()) })?;
    Ok(out.chunks_exact(18).map(g1_from_jacobian_limbs).collect())
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

/// Hook for the optional patched ark-ec (vendor/ark-ec-hip/msm_hip.patch adds a run-time hook REGISTRY to ark-ec; ark-ec
/// cannot depend on this crate -- ark-bls12-381, which the leaf crate needs, depends on ark-ec).  [`install_msm_hook`]
/// registers [`msm_hook_erased`].  The hook answers only for base slices that lie inside a committer key registered by
/// `PC::trim` of the wrappers (`SrsCache::resolve_containing`: no upload, no table build, no device state for arbitrary
/// slices), checks the slice's end points against the device copy, and declines -- `false`, ark-ec's own Pippenger runs --
/// for any other curve, for short MSMs, for unknown slices, for a mismatch, or on a library error.  `scalars` are canonical
/// `BigInteger256`s (what `multi_scalar_mul` receives), passed with `scalars_are_montgomery = 0`.  The identity cannot be
/// an uploaded base (`GpuSrs::upload` refuses it), so registered slices contain none.
pub fn msm_hook(bases: &[G1Affine], scalars: &[[u64; 4]]) -> Option<G1Projective> {
    let n = core::cmp::min(bases.len(), scalars.len());
    if n < GPU_MSM_THRESHOLD {
        return None;
    }
    let bases = &bases[..n];
    let (srs, offset) = srs_cache().resolve_containing(bases)?;
    // the device copy must still be this memory's image: compare the first and the last point of the slice
    let mut ends = [0u64; 24];
    check(unsafe { ffi::mh_bases_download(srs.handle, offset, 1, ends.as_mut_ptr()) }).ok()?;
    check(unsafe { ffi::mh_bases_download(srs.handle, offset + n - 1, 1, ends[12..].as_mut_ptr()) }).ok()?;
    let want = crate::convert::g1_slice_to_limbs_checked(&[bases[0], bases[n - 1]])?;
    if want[..] != ends[..] {
        return None;
    }
    let mut limbs = Vec::with_capacity(n * 4);
    for s in &scalars[..n] {
        limbs.extend_from_slice(s);
    }
    let mut out = [0u64; 18];
    check(unsafe { ffi::mh_msm(srs.handle, offset, limbs.as_ptr(), 0, n, out.as_mut_ptr()) }).ok()?;
    Some(g1_from_jacobian_limbs(&out))
}

/// The type-erased form the patched ark-ec calls: `g` identifies `G` (only `ark_bls12_381::G1Affine` is served), `bases` /
/// `scalars` point at `n_bases` `G` values / `n_scalars` `BigInteger256`s, `out` at an uninitialised `G::Projective`.
pub fn msm_hook_erased(g: std::any::TypeId, bases: *const u8, n_bases: usize, scalars: *const u64, n_scalars: usize, out: *mut u8) -> bool {
    if g != std::any::TypeId::of::<G1Affine>() {
        return false;
    }
    // SAFETY: the patched `multi_scalar_mul::<G>` passes its own argument slices and `G == G1Affine` was just checked;
    // `BigInteger256` is a newtype of `[u64; 4]`
    let b: &[G1Affine] = unsafe { core::slice::from_raw_parts(bases as *const G1Affine, n_bases) };
    let s: &[[u64; 4]] = unsafe { core::slice::from_raw_parts(scalars as *const [u64; 4], n_scalars) };
    match msm_hook(b, s) {
        Some(p) => {
            unsafe { core::ptr::write(out as *mut G1Projective, p) };
            true
        }
        None => false,
    }
}

/// Registers the hook with the patched ark-ec (a no-op route unless `[patch.crates-io] ark-ec` is enabled in Cargo.toml).
#[cfg(feature = "ark-ec-hook")]
pub fn install_msm_hook() {
    ensure_init();
    ark_ec::msm::set_multi_scalar_mul_hook(Some(msm_hook_erased));
}
