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

pub mod kzg;
pub mod marlin_pc;
pub mod prover;
pub mod sonic_pc;

// the leaf crate: raw bindings, marshalling, the ark-poly hook (kept there so that the patched ark-poly does not depend on
// this crate, which depends on ark-poly)
pub use marlin_hip_sys::{check, convert, ensure_init, ffi, ntt, HipError};

pub use kzg::{msm_g1, GpuSrs};
