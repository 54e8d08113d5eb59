//! Marshalling between arkworks 0.3 values and the u64-limb layout of `include/marlin_hip.h`.
//!
//! `Fp256(BigInteger256([u64; 4]), PhantomData)` and `GroupAffine { x, y, infinity, .. }` are `repr(Rust)`:
//! values are copied limb by limb, never transmuted.  The inner `BigInteger` of an `Fp*` IS the
//! Montgomery representation, which is what the library computes in, so no conversion happens on either side.
//!
//! UNCOMPILED (see Cargo.toml).
use ark_bls12_381::{Fq, Fr, G1Affine, G1Projective};
use ark_ff::{BigInteger256, BigInteger384, Fp256, Fp384};

pub const FR_LIMBS: usize = 4;
pub const FQ_LIMBS: usize = 6;

/// Montgomery limbs of scalars, 4 x u64 each (pass with `scalars_are_montgomery = 1`).
pub fn fr_slice_to_limbs(v: &[Fr]) -> Vec<u64> {
    let mut out = Vec::with_capacity(v.len() * FR_LIMBS);
    for x in v {
        out.extend_from_slice(&(x.0).0);
    }
    out
}

/// Inverse of [`fr_slice_to_limbs`]: `Fp256::new` wraps a raw (Montgomery) `BigInteger256`.
pub fn limbs_to_fr_slice(limbs: &[u64]) -> Vec<Fr> {
    limbs
        .chunks_exact(FR_LIMBS)
        .map(|c| Fp256::new(BigInteger256([c[0], c[1], c[2], c[3]])))
        .collect()
}

pub fn fq_from_mont(l: &[u64]) -> Fq {
    Fp384::new(BigInteger384([l[0], l[1], l[2], l[3], l[4], l[5]]))
}

/// x || y Montgomery limbs of affine points (12 x u64 each).  The boundary has no infinity flag
/// (marlin_hip.h:22-24): an SRS never contains the identity, and `mh_bases_upload` rejects off-curve encodings.
pub fn g1_slice_to_limbs(v: &[G1Affine]) -> Vec<u64> {
    let mut out = Vec::with_capacity(v.len() * 2 * FQ_LIMBS);
    for p in v {
        assert!(!p.infinity, "the identity cannot be an MSM base at this boundary");
        out.extend_from_slice(&(p.x.0).0);
        out.extend_from_slice(&(p.y.0).0);
    }
    out
}

/// The same, `None` when a base is the identity (legal for `VariableBaseMSM`, not expressible at this boundary: the
/// caller then keeps the host path instead of panicking).
pub fn g1_slice_to_limbs_checked(v: &[G1Affine]) -> Option<Vec<u64>> {
    if v.iter().any(|p| p.infinity) {
        return None;
    }
    Some(g1_slice_to_limbs(v))
}

/// Jacobian X || Y || Z (18 limbs, Z = 0 for the identity) -> `G1Projective`.
pub fn g1_from_jacobian_limbs(l: &[u64]) -> G1Projective {
    G1Projective::new(fq_from_mont(&l[0..6]), fq_from_mont(&l[6..12]), fq_from_mont(&l[12..18]))
}
