//! Raw bindings: one declaration per prototype of `include/marlin_hip.h`, in the header's order.
//! `tests/test_capi_symbols.py` (Python side of this repository) checks header <-> exported symbols;
//! `tests/ffi_symbols.rs` takes the address of every item below so that a missing symbol is a
//! link error, not a run-time surprise.
//!
//! UNCOMPILED (no Rust toolchain in the development image) -- see Cargo.toml.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int, c_void};

pub const MH_OK: c_int = 0;
pub const MH_EINVAL: c_int = -1;
pub const MH_ENOMEM: c_int = -2;
pub const MH_EHIP: c_int = -3;
pub const MH_ENOINIT: c_int = -4;
pub const MH_ENODEV: c_int = -5;
pub const MH_ECHECK: c_int = -6;

/// `mh_ctx_t` / `mh_group_t` (marlin_hip.h): opaque handles of a context (one GPU's worth of library state) and of a group of
/// contexts joined by the in-process transport.
#[repr(C)]
pub struct mh_context { _private: [u8; 0] }
pub type mh_ctx_t = *mut mh_context;
#[repr(C)]
pub struct mh_group { _private: [u8; 0] }
pub type mh_group_t = *mut mh_group;
pub type mh_group_fn = Option<unsafe extern "C" fn(rank: c_int, user: *mut c_void) -> c_int>;

pub const MH_FIELD_BLS12_381_FR: c_int = 0;
pub const MH_CURVE_BLS12_381_G1: c_int = 0;
pub const MH_FIELD_BN254_FR: c_int = 1;
pub const MH_CURVE_BN254_G1: c_int = 1;

/// `mh_r1cs_matrices` (marlin_hip.h): the padded square R1CS as CSR, A / B / C.
#[repr(C)]
pub struct mh_r1cs_matrices {
    pub num_constraints: u64,
    pub num_instance: u64,
    pub row_ptr: [*const u64; 3],
    pub col: [*const u32; 3],
    pub val: [*const u64; 3],
}

/// `mh_verifier_key` (marlin_hip.h): the group elements of kzg10::VerifierKey and marlin_pc:: / sonic_pc::VerifierKey
/// (the two shift powers are G1 points for MarlinKZG10, G2 points for SonicKZG10).
#[repr(C)]
pub struct mh_verifier_key {
    pub g_xy: *const u64,
    pub gamma_g_xy: *const u64,
    pub h_xy: *const u64,
    pub beta_h_xy: *const u64,
    pub shift_power_h_xy: *const u64,
    pub shift_power_k_xy: *const u64,
}

/// `mh_fiat_shamir`: the caller's `FS: FiatShamirRng` as three callbacks (`mh_marlin_prove_fs` / `mh_marlin_verify_fs`).
#[repr(C)]
pub struct mh_fiat_shamir {
    pub user: *mut c_void,
    pub initialize: Option<unsafe extern "C" fn(user: *mut c_void, input: *const u8, len: usize)>,
    pub absorb: Option<unsafe extern "C" fn(user: *mut c_void, input: *const u8, len: usize)>,
    pub next_u64: Option<unsafe extern "C" fn(user: *mut c_void) -> u64>,
}

/// `mh_allgather_fn`: gather `bytes` bytes from every rank into `recv` (rank-major).
pub type mh_allgather_fn =
    Option<unsafe extern "C" fn(send: *const c_void, bytes: usize, recv: *mut c_void, user: *mut c_void) -> c_int>;

/// `mh_alltoall_fn`: all-to-all on DEVICE buffers (chunk q of `d_send` to rank q, chunk q of `d_recv` from rank q).
pub type mh_alltoall_fn =
    Option<unsafe extern "C" fn(d_send: *const c_void, bytes_per_peer: usize, d_recv: *mut c_void, user: *mut c_void) -> c_int>;

/// `mh_allgather_dev_fn`: all-gather on DEVICE buffers (`bytes` from every rank into `d_recv`, rank-major).
pub type mh_allgather_dev_fn =
    Option<unsafe extern "C" fn(d_send: *const c_void, bytes: usize, d_recv: *mut c_void, user: *mut c_void) -> c_int>;

extern "C" {
    pub fn mh_curve_info(curve_id: *mut c_int, fr_limbs64: *mut c_int, fq_limbs64: *mut c_int, fr_two_adicity: *mut c_int) -> c_int;

    // ---- lifecycle
    pub fn mh_init(device_id: c_int) -> c_int;
    pub fn mh_init_devices(device_ids: *const c_int, n_devices: c_int) -> c_int;
    pub fn mh_shutdown() -> c_int;
    pub fn mh_last_error() -> *const c_char;
    pub fn mh_set_stream(hip_stream: *mut c_void) -> c_int;
    pub fn mh_synchronize() -> c_int;
    pub fn mh_device_info(name_out: *mut c_char, name_cap: usize, cu_count: *mut c_int, hbm_bytes: *mut usize) -> c_int;
    pub fn mh_ctx_create(device_id: c_int, ctx_out: *mut mh_ctx_t) -> c_int;
    pub fn mh_ctx_set_current(ctx: mh_ctx_t) -> c_int;
    pub fn mh_ctx_get_current() -> mh_ctx_t;
    pub fn mh_ctx_destroy(ctx: mh_ctx_t) -> c_int;
    pub fn mh_group_create(device_ids: *const c_int, world: c_int, group_out: *mut mh_group_t) -> c_int;
    pub fn mh_group_size(group: mh_group_t) -> c_int;
    pub fn mh_group_ctx(group: mh_group_t, rank: c_int) -> mh_ctx_t;
    pub fn mh_group_run(group: mh_group_t, f: mh_group_fn, user: *mut c_void) -> c_int;
    pub fn mh_group_destroy(group: mh_group_t) -> c_int;

    // ---- device memory
    pub fn mh_alloc(bytes: usize, dptr_out: *mut *mut c_void) -> c_int;
    pub fn mh_free(dptr: *mut c_void) -> c_int;
    pub fn mh_memcpy_h2d(dst_dev: *mut c_void, src_host: *const c_void, bytes: usize) -> c_int;
    pub fn mh_memcpy_d2h(dst_host: *mut c_void, src_dev: *const c_void, bytes: usize) -> c_int;
    pub fn mh_memcpy_d2d(dst_dev: *mut c_void, src_dev: *const c_void, bytes: usize) -> c_int;
    pub fn mh_memset(dst_dev: *mut c_void, byte: c_int, bytes: usize) -> c_int;

    // ---- NTT over Fr (seam B2)
    pub fn mh_ntt(field: c_int, data_mont: *mut u64, log_n: u32, inverse: c_int) -> c_int;
    pub fn mh_ntt_dev(field: c_int, d_in: *const c_void, d_out: *mut c_void, log_n: u32, inverse: c_int) -> c_int;
    pub fn mh_ntt_len(field: c_int, data_mont: *mut u64, in_len: usize, log_n: u32, inverse: c_int) -> c_int;
    pub fn mh_ntt_coset(field: c_int, data_mont: *mut u64, log_n: u32, inverse: c_int) -> c_int;
    pub fn mh_ntt_coset_dev(field: c_int, d_in: *const c_void, d_out: *mut c_void, log_n: u32, inverse: c_int) -> c_int;

    // ---- MSM over G1 (seam B1)
    pub fn mh_bases_upload(curve: c_int, xy_mont: *const u64, n: usize, handle_out: *mut u64) -> c_int;
    pub fn mh_bases_from_dev(curve: c_int, d_xy_mont: *const c_void, n: usize, handle_out: *mut u64) -> c_int;
    pub fn mh_bases_upload_serialized(curve: c_int, bytes: *const u8, n: usize, compressed: c_int, handle_out: *mut u64) -> c_int;
    pub fn mh_srs_powers(curve: c_int, tau_mont: *const u64, scale_mont: *const u64, first: usize, n: usize, handle_out: *mut u64) -> c_int;
    pub fn mh_bases_download(handle: u64, offset: usize, n: usize, xy_mont_out: *mut u64) -> c_int;
    pub fn mh_bases_free(handle: u64) -> c_int;
    pub fn mh_bases_len(handle: u64, n_out: *mut usize) -> c_int;
    pub fn mh_bases_precompute(handle: u64, window_bits: u32) -> c_int;
    pub fn mh_bases_table_info(handle: u64, window_bits: *mut u32, windows: *mut u32, table_bytes: *mut u64) -> c_int;
    pub fn mh_msm_path_counts(fixed_base_groups: *mut u64, variable_base_groups: *mut u64) -> c_int;
    pub fn mh_msm(bases_handle: u64, base_offset: usize, scalars: *const u64, scalars_are_montgomery: c_int, n: usize, out_xyz_mont: *mut u64) -> c_int;
    pub fn mh_msm_dev(bases_handle: u64, base_offset: usize, d_scalars: *const c_void, scalars_are_montgomery: c_int, n: usize, out_xyz_mont: *mut u64) -> c_int;
    pub fn mh_msm_batch_dev(njobs: usize, bases_handles: *const u64, base_offsets: *const usize, d_scalars: *const *const c_void,
                            ns: *const usize, scalars_are_montgomery: c_int, out_xyz_mont: *mut u64) -> c_int;
    pub fn mh_msm_batch(njobs: usize, bases_handles: *const u64, base_offsets: *const usize, scalars: *const *const u64,
                        ns: *const usize, scalars_are_montgomery: c_int, out_xyz_mont: *mut u64) -> c_int;
    pub fn mh_msm_batch_sharded_dev(njobs: usize, bases_handles: *const u64, base_offsets: *const usize, d_scalars: *const *const c_void,
                                    ns: *const usize, scalars_are_montgomery: c_int, out_xyz_mont: *mut u64) -> c_int;
    pub fn mh_g1_to_affine(xyz_mont: *const u64, xy_mont_out: *mut u64, is_infinity_out: *mut c_int) -> c_int;
    pub fn mh_g1_sum(xyz_points: *const u64, n: usize, out_xyz: *mut u64) -> c_int;

    // ---- G2 (verifier side of the SRS)
    pub fn mh_g2_bases_upload(curve: c_int, xy_mont: *const u64, n: usize, handle_out: *mut u64) -> c_int;
    pub fn mh_g2_srs_powers(curve: c_int, gen_xy_mont: *const u64, tau_mont: *const u64, scale_mont: *const u64, first: usize, n: usize,
                            handle_out: *mut u64) -> c_int;
    pub fn mh_g2_bases_download(handle: u64, offset: usize, n: usize, xy_mont_out: *mut u64) -> c_int;
    pub fn mh_g2_bases_free(handle: u64) -> c_int;
    pub fn mh_g2_msm(handle: u64, base_offset: usize, scalars: *const u64, scalars_are_montgomery: c_int, n: usize,
                     out_xy_mont: *mut u64, is_infinity_out: *mut c_int) -> c_int;

    // ---- Marlin index / prove with device-resident polynomials
    pub fn mh_marlin_index(m: *const mh_r1cs_matrices, srs_g: u64, srs_gamma_g: u64, pk_out: *mut u64) -> c_int;
    pub fn mh_marlin_index_pc(m: *const mh_r1cs_matrices, srs_g: u64, srs_gamma_g: u64, pc: c_int, pk_out: *mut u64) -> c_int;
    pub fn mh_marlin_pk_free(pk: u64) -> c_int;
    pub fn mh_marlin_pk_info(pk: u64, info8: *mut u64) -> c_int;
    pub fn mh_marlin_vk_bytes(pk: u64, out: *mut u8, cap: usize, len_out: *mut usize) -> c_int;
    pub fn mh_marlin_prove(pk: u64, instance_mont: *const u64, witness_mont: *const u64, zk_seed32: *const u8,
                           zk_chacha_rounds: c_int, proof_out: *mut u8, cap: usize, len_out: *mut usize) -> c_int;
    pub fn mh_marlin_prove_dev(pk: u64, d_instance_mont: *const c_void, d_witness_mont: *const c_void, zk_seed32: *const u8,
                               zk_chacha_rounds: c_int, proof_out: *mut u8, cap: usize, len_out: *mut usize) -> c_int;
    pub fn mh_marlin_prove_fs(pk: u64, instance_mont: *const u64, witness_mont: *const u64, zk_seed32: *const u8, zk_chacha_rounds: c_int,
                              fs: *const mh_fiat_shamir, proof_out: *mut u8, cap: usize, len_out: *mut usize) -> c_int;
    pub fn mh_marlin_verify_fs(vk_bytes: *const u8, vk_len: usize, vk: *const mh_verifier_key, pc: c_int, public_input_mont: *const u64, n_public: usize,
                               flat_proof: *const u8, proof_len: usize, fs: *const mh_fiat_shamir, ok_out: *mut c_int) -> c_int;
    pub fn mh_marlin_zk_draw_count(pk: u64, n_out: *mut usize) -> c_int;
    pub fn mh_marlin_prove_draws(pk: u64, instance_mont: *const u64, witness_mont: *const u64, zk_draws_mont: *const u64, n_draws: usize,
                                 proof_out: *mut u8, cap: usize, len_out: *mut usize) -> c_int;
    pub fn mh_marlin_verify(vk_bytes: *const u8, vk_len: usize, vk: *const mh_verifier_key, pc: c_int, public_input_mont: *const u64, n_public: usize,
                            flat_proof: *const u8, proof_len: usize, ok_out: *mut c_int) -> c_int;
    pub fn mh_pairing_product_is_one(g1_xy_mont: *const u64, g2_xy_mont: *const u64, n: usize, is_one_out: *mut c_int) -> c_int;
    pub fn mh_marlin_proof_serialize(flat_proof: *const u8, flat_len: usize, pc: c_int, out: *mut u8, cap: usize, len_out: *mut usize) -> c_int;
    pub fn mh_marlin_proof_deserialize(bytes: *const u8, len: usize, pc: c_int, flat_out: *mut u8, cap: usize, len_out: *mut usize) -> c_int;
    pub fn mh_marlin_set_shard(rank: c_int, world: c_int, allgather: mh_allgather_fn, user: *mut c_void) -> c_int;
    pub fn mh_marlin_set_alltoall(alltoall: mh_alltoall_fn, user: *mut c_void) -> c_int;
    pub fn mh_marlin_set_alltoall_mode(stream_ordered: c_int) -> c_int;
    pub fn mh_marlin_set_allgather_dev(allgather_dev: mh_allgather_dev_fn, user: *mut c_void) -> c_int;
    pub fn mh_marlin_probe_exchange_dev(which: c_int, d_send: *const c_void, bytes: usize, d_recv: *mut c_void) -> c_int;
    // native transport: RCCL called by the library on its own stream (marlin_amd/csrc/rccl_native.h)
    pub fn mh_rccl_unique_id(id128_out: *mut u8) -> c_int;
    pub fn mh_marlin_set_rccl(rank: c_int, world: c_int, id128: *const u8) -> c_int;
    pub fn mh_marlin_rccl_sliced(sliced: c_int) -> c_int;
    pub fn mh_marlin_rccl_destroy() -> c_int;
    pub fn mh_marlin_rccl_info(info4: *mut u64, lib_path: *mut c_char, cap: usize) -> c_int;
    pub fn mh_marlin_exchange_stats(calls_out: *mut u64, host_ms_out: *mut f64, reset: c_int) -> c_int;
    pub fn mh_ntt_dist_dev(field: c_int, d_in_local: *const c_void, d_out_local: *mut c_void, log_n: u32, inverse: c_int) -> c_int;
    pub fn mh_msm_batch_sliced_dev(bases_handle: u64, njobs: usize, first_index: *const usize, stride: usize, d_scalars_local: *const *const c_void,
                                   ns_local: *const usize, scalars_are_montgomery: c_int, combine: c_int, out_xyz_mont: *mut u64) -> c_int;
    pub fn mh_marlin_probe_allgather(send: *const c_void, bytes: usize, recv: *mut c_void) -> c_int;
    pub fn mh_marlin_get_poly(pk: u64, label: *const c_char, out: *mut u64, cap_elems: usize, len_out: *mut usize) -> c_int;

    // ---- profiling / self-test
    pub fn mh_prof_enable(on: c_int) -> c_int;
    pub fn mh_prof_reset() -> c_int;
    pub fn mh_prof_get(family: c_int, total_ms_out: *mut f64, launches_out: *mut u64) -> c_int;
    pub fn mh_check_level(level: c_int) -> c_int;
    pub fn mh_check_report(out: *mut c_char, cap: usize, counts2: *mut u64) -> c_int;
}

/// The library's thread-local message for the last failure on this thread.
pub fn last_error() -> String {
    unsafe {
        let p = mh_last_error();
        if p.is_null() {
            String::new()
        } else {
            std::ffi::CStr::from_ptr(p).to_string_lossy().into_owned()
        }
    }
}
