//! marlin-hip: the Rust side of the drop-in boundary of this repository.
//!
//! UNCOMPILED -- written without a Rust toolchain (see Cargo.toml).  Three routes, smallest first:
//!
//! * [`msm_g1`] / [`GpuSrs`]: `VariableBaseMSM::multi_scalar_mul` over G1 -> `mh_msm`.
//! * seam B1 (`/root/reference/src/lib.rs:64,70`): [`marlin_pc::GpuMarlinKZG10`] and
//!   [`sonic_pc::GpuSonicKZG10`] are `PolynomialCommitment`s that reuse every upstream key / commitment /
//!   proof type and replace only the multi-scalar multiplications inside `commit` and `open`:
//!   `type MultiPC = GpuMarlinKZG10<Bls12_381, DensePolynomial<Fr>>` is the one-word change to
//!   `src/test.rs:123`.
//! * seam B2 (`src/ahp/prover.rs:49-55`): the patched `ark-poly` of `vendor/ark-poly-hip/` calls
//!   [`ntt::fft_in_place_hook`].
//! * the whole-prover route ([`prover::GpuMarlin`]): `Marlin::{index, prove}` with device-resident
//!   polynomials (`mh_marlin_index`, `mh_marlin_prove`); what `bench.py` measures.
//!
//! `tests/parity.rs` runs the reference's five `src/test.rs` shapes on the stock stack and on each route and
//! asserts equal `CanonicalSerialize` bytes -- the check that pins this repository's oracle to arkworks.

pub mod convert;
pub mod ffi;
pub mod kzg;
pub mod marlin_pc;
pub mod ntt;
pub mod prover;
pub mod sonic_pc;

use std::os::raw::c_int;
use std::sync::Once;

/// Error of every route; `GpuMarlinKZG10::Error` is upstream's `ark_poly_commit::Error`, into which this
/// converts as `Error::IncorrectInputLength`-style string errors do upstream.
#[derive(Debug)]
pub enum HipError {
    /// `MH_E*` code and the library's message (`mh_last_error`).
    Library(c_int, String),
    /// The boundary cannot express the request (for instance a `zk_rng` that is not a ChaCha generator).
    Unsupported(&'static str),
}

impl core::fmt::Display for HipError {
    fn fmt(&self, f: &mut core::fmt::Formatter<'_>) -> core::fmt::Result {
        match self {
            HipError::Library(rc, msg) => write!(f, "libmarlin_hip: error {}: {}", rc, msg),
            HipError::Unsupported(what) => write!(f, "marlin-hip: unsupported: {}", what),
        }
    }
}
impl std::error::Error for HipError {}

/// Maps a C status to `Result` (every entry point returns 0 or a negative `MH_E*`, never unwinds).
pub(crate) fn check(rc: c_int) -> Result<(), HipError> {
    if rc == ffi::MH_OK {
        Ok(())
    } else {
        Err(HipError::Library(rc, ffi::last_error()))
    }
}

static INIT: Once = Once::new();

/// `mh_init(MARLIN_HIP_DEVICE or 0)`, once per process (one process drives one GPU; marlin_hip.h:57-62).
pub fn ensure_init() {
    INIT.call_once(|| {
        let dev = std::env::var("MARLIN_HIP_DEVICE").ok().and_then(|s| s.parse().ok()).unwrap_or(0);
        let rc = unsafe { ffi::mh_init(dev) };
        assert_eq!(rc, 0, "mh_init({}) failed: {}", dev, ffi::last_error());
    });
}

pub use kzg::{msm_g1, GpuSrs};
