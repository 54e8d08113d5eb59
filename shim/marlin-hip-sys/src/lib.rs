//! marlin-hip-sys: raw bindings of `include/marlin_hip.h` ([`ffi`]), marshalling between arkworks 0.3 values and the
//! library's u64-limb layout ([`convert`]), and the hook the patched `ark-poly` calls from its radix-2 domain
//! ([`ntt`], seam B2 of SURVEY.md 8b).  A leaf: it depends on `ark-ff` and the curve crate only, so the patched
//! `ark-poly` can depend on it without a cycle (`marlin-hip` depends on both; see Cargo.toml).
//!
//! UNCOMPILED -- written without a Rust toolchain.
pub mod convert;
pub mod ffi;
pub mod ntt;

use std::os::raw::c_int;
use std::sync::Once;

/// Error of every route; `GpuMarlinKZG10::Error` is upstream's `ark_poly_commit::Error`, into which this converts.
#[derive(Debug)]
pub enum HipError {
    /// `MH_E*` code and the library's message (`mh_last_error`).
    Library(c_int, String),
    /// The boundary cannot express the request.
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
pub fn check(rc: c_int) -> Result<(), HipError> {
    if rc == ffi::MH_OK {
        Ok(())
    } else {
        Err(HipError::Library(rc, ffi::last_error()))
    }
}

static INIT: Once = Once::new();

/// `mh_init(MARLIN_HIP_DEVICE or 0)`, once per process (one process drives one GPU).
pub fn ensure_init() {
    INIT.call_once(|| {
        let dev = std::env::var("MARLIN_HIP_DEVICE").ok().and_then(|s| s.parse().ok()).unwrap_or(0);
        let rc = unsafe { ffi::mh_init(dev) };
        assert_eq!(rc, 0, "mh_init({}) failed: {}", dev, ffi::last_error());
    });
}
