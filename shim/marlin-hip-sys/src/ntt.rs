//! Seam B2: the hooks the patched `ark-poly` (vendor/ark-poly-hip/radix2_hip.patch) calls from
//! `Radix2EvaluationDomain::{fft_in_place, ifft_in_place, coset_fft_in_place, coset_ifft_in_place}`.
//!
//! `GeneralEvaluationDomain<F>` is a concrete type inside `ProverState` (`/root/reference/src/ahp/prover.rs:49-55`,
//! constructed at `:280-287`), so the evaluation domain cannot be swapped by a type parameter; a `[patch.crates-io]`
//! fork is the only non-invasive route.  Semantics preserved: the caller has already zero-padded to
//! the domain size; natural order in and out; the inverse multiplies by n^-1; coset variants multiply by powers of
//! `F::multiplicative_generator()` (7 for BLS12-381 Fr) before / after.
//!
//! Lives in the leaf crate so that the patched `ark-poly` depends on `marlin-hip-sys` only (no cycle through
//! `marlin-hip` -> `ark-poly`).
//!
//! UNCOMPILED (see Cargo.toml).
use crate::convert::{fr_slice_to_limbs, limbs_to_fr_slice};
use crate::{check, ensure_init, ffi};
use ark_bls12_381::Fr;
use std::any::TypeId;

/// Transforms below 2^12 stay on the host: PCIe + launch latency exceeds the host's own time there.
pub const GPU_NTT_THRESHOLD_LOG: u32 = 12;

#[derive(Clone, Copy)]
pub enum Kind {
    Fft,
    Ifft,
    CosetFft,
    CosetIfft,
}

/// Returns `true` when the transform was done on the device (the caller then skips its own butterflies).
/// `T` is ark-poly's `DomainCoeff<F>`; only `T = F = ark_bls12_381::Fr` is accelerated (group-element coefficient
/// vectors, which `DomainCoeff` also admits, keep the host path).
/// `in_len`: how many elements the caller's `Vec` held BEFORE ark-poly padded it to the domain size (`coeffs.resize(self.size(),
/// T::zero())` at the top of `fft_in_place`): only those are uploaded (`mh_ntt_len`); most forward transforms of Marlin's prover
/// are of polynomials much shorter than their domain (H + 1 coefficients on 4H points).
pub fn fft_in_place_hook<T: 'static>(coeffs: &mut [T], in_len: usize, log_size_of_group: u32, kind: Kind) -> bool {
    if TypeId::of::<T>() != TypeId::of::<Fr>() || log_size_of_group < GPU_NTT_THRESHOLD_LOG {
        return false;
    }
    debug_assert_eq!(coeffs.len(), 1usize << log_size_of_group);
    // SAFETY: T == Fr was just checked; the slice is reinterpreted, not the elements' layout.
    let fr: &mut [Fr] = unsafe { core::slice::from_raw_parts_mut(coeffs.as_mut_ptr() as *mut Fr, coeffs.len()) };
    ensure_init();
    let mut limbs = fr_slice_to_limbs(fr);
    let (coset, inverse) = match kind {
        Kind::Fft => (false, 0),
        Kind::Ifft => (false, 1),
        Kind::CosetFft => (true, 0),
        Kind::CosetIfft => (true, 1),
    };
    let rc = unsafe {
        if coset {
            ffi::mh_ntt_coset(ffi::MH_FIELD_BLS12_381_FR, limbs.as_mut_ptr(), log_size_of_group, inverse)
        } else {
            ffi::mh_ntt_len(ffi::MH_FIELD_BLS12_381_FR, limbs.as_mut_ptr(), in_len.min(1usize << log_size_of_group), log_size_of_group, inverse)
        }
    };
    if check(rc).is_err() {
        return false; // the host path still produces the same values
    }
    fr.copy_from_slice(&limbs_to_fr_slice(&limbs));
    true
}
