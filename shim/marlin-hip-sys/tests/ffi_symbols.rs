//! Link-time check: the address of every binding of src/ffi.rs is taken, so a symbol missing from
//! libmarlin_hip.so fails the build of this test.  UNCOMPILED (see Cargo.toml).
use marlin_hip_sys::ffi::*;

#[test]
fn every_header_symbol_links() {
    let addrs: &[usize] = &[
        mh_curve_info as usize, mh_init as usize, mh_init_devices as usize, mh_shutdown as usize, mh_last_error as usize,
        mh_set_stream as usize, mh_synchronize as usize, mh_device_info as usize, mh_alloc as usize, mh_free as usize,
        mh_memcpy_h2d as usize, mh_memcpy_d2h as usize, mh_memcpy_d2d as usize, mh_memset as usize, mh_ntt as usize,
        mh_ntt_dev as usize, mh_ntt_len as usize, mh_ntt_coset as usize, mh_ntt_coset_dev as usize, mh_bases_upload as usize,
        mh_bases_from_dev as usize, mh_bases_upload_serialized as usize, mh_srs_powers as usize, mh_bases_download as usize, mh_bases_free as usize,
        mh_bases_len as usize, mh_bases_precompute as usize, mh_bases_table_info as usize, mh_msm_path_counts as usize,
        mh_msm as usize, mh_msm_dev as usize, mh_msm_batch_dev as usize, mh_msm_batch as usize, mh_msm_batch_sharded_dev as usize, mh_g1_to_affine as usize, mh_g1_sum as usize,
        mh_marlin_index as usize, mh_marlin_index_pc as usize, mh_marlin_pk_free as usize, mh_marlin_pk_info as usize,
        mh_marlin_vk_bytes as usize, mh_marlin_prove as usize, mh_marlin_prove_dev as usize, mh_marlin_prove_fs as usize, mh_marlin_verify_fs as usize, mh_marlin_zk_draw_count as usize, mh_marlin_prove_draws as usize, mh_marlin_verify as usize, mh_pairing_product_is_one as usize, mh_marlin_proof_serialize as usize,
        mh_marlin_proof_deserialize as usize, mh_marlin_set_shard as usize, mh_marlin_probe_allgather as usize, mh_marlin_set_alltoall as usize, mh_marlin_set_alltoall_mode as usize, mh_marlin_set_allgather_dev as usize, mh_marlin_probe_exchange_dev as usize, mh_rccl_unique_id as usize, mh_marlin_set_rccl as usize, mh_marlin_rccl_sliced as usize, mh_marlin_rccl_destroy as usize, mh_marlin_rccl_info as usize, mh_marlin_exchange_stats as usize,
        mh_ntt_dist_dev as usize, mh_msm_batch_sliced_dev as usize,
        mh_marlin_get_poly as usize, mh_prof_enable as usize, mh_prof_reset as usize, mh_prof_get as usize,
        mh_g2_bases_upload as usize, mh_g2_srs_powers as usize, mh_g2_bases_download as usize,
        mh_g2_bases_free as usize, mh_g2_msm as usize, mh_check_level as usize, mh_check_report as usize,
        mh_ctx_create as usize, mh_ctx_set_current as usize, mh_ctx_get_current as usize, mh_ctx_destroy as usize,
        mh_group_create as usize, mh_group_size as usize, mh_group_ctx as usize, mh_group_run as usize, mh_group_destroy as usize,
    ];
    assert!(addrs.iter().all(|a| *a != 0));
}
