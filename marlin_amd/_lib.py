"""ctypes binding of libmarlin_hip.so (the C ABI declared in include/marlin_hip.h).

There is no CPU fallback: if the shared library is missing or a GPU call fails,
the error propagates.  The oracle under /oracle is never imported from here.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# one library per curve (build-time choice, same ABI); the process picks one with MARLIN_AMD_CURVE
CURVE = os.environ.get("MARLIN_AMD_CURVE", "bls12_381")
LIB_PATH = os.path.join(_HERE, {"bls12_381": "libmarlin_hip.so", "bn254": "libmarlin_hip_bn254.so"}[CURVE])
# A/B measurements of two builds on one box: MARLIN_AMD_LIB=<path to another build of the same ABI>
LIB_PATH = os.environ.get("MARLIN_AMD_LIB", LIB_PATH)

# every symbol include/marlin_hip.h declares: (restype, argtypes)
_u64p = C.POINTER(C.c_uint64)
SYMBOLS = {
    "mh_init": (C.c_int, [C.c_int]),
    "mh_init_devices": (C.c_int, [C.POINTER(C.c_int), C.c_int]),
    "mh_shutdown": (C.c_int, []),
    "mh_last_error": (C.c_char_p, []),
    "mh_set_stream": (C.c_int, [C.c_void_p]),
    "mh_synchronize": (C.c_int, []),
    "mh_device_info": (C.c_int, [C.c_char_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_size_t)]),
    "mh_ctx_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "mh_ctx_set_current": (C.c_int, [C.c_void_p]),
    "mh_ctx_get_current": (C.c_void_p, []),
    "mh_ctx_destroy": (C.c_int, [C.c_void_p]),
    "mh_group_create": (C.c_int, [C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)]),
    "mh_group_size": (C.c_int, [C.c_void_p]),
    "mh_group_ctx": (C.c_void_p, [C.c_void_p, C.c_int]),
    "mh_group_run": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "mh_group_destroy": (C.c_int, [C.c_void_p]),
    "mh_curve_info": (C.c_int, [C.POINTER(C.c_int)] * 4),
    "mh_alloc": (C.c_int, [C.c_size_t, C.POINTER(C.c_void_p)]),
    "mh_free": (C.c_int, [C.c_void_p]),
    "mh_memcpy_h2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "mh_memcpy_d2h": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "mh_memcpy_d2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "mh_memset": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t]),
    "mh_ntt": (C.c_int, [C.c_int, C.c_void_p, C.c_uint32, C.c_int]),
    "mh_ntt_len": (C.c_int, [C.c_int, C.c_void_p, C.c_size_t, C.c_uint32, C.c_int]),
    "mh_ntt_dev": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int]),
    "mh_ntt_coset": (C.c_int, [C.c_int, C.c_void_p, C.c_uint32, C.c_int]),
    "mh_ntt_coset_dev": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int]),
    "mh_bases_upload": (C.c_int, [C.c_int, C.c_void_p, C.c_size_t, _u64p]),
    "mh_bases_upload_serialized": (C.c_int, [C.c_int, C.c_char_p, C.c_size_t, C.c_int, C.POINTER(C.c_uint64)]),
    "mh_bases_from_dev": (C.c_int, [C.c_int, C.c_void_p, C.c_size_t, _u64p]),
    "mh_srs_powers": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, _u64p]),
    "mh_bases_download": (C.c_int, [C.c_uint64, C.c_size_t, C.c_size_t, C.c_void_p]),
    "mh_bases_free": (C.c_int, [C.c_uint64]),
    "mh_bases_len": (C.c_int, [C.c_uint64, C.POINTER(C.c_size_t)]),
    "mh_bases_precompute": (C.c_int, [C.c_uint64, C.c_uint32]),
    "mh_msm_path_counts": (C.c_int, [_u64p, _u64p]),
    "mh_bases_table_info": (C.c_int, [C.c_uint64, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), _u64p]),
    "mh_msm": (C.c_int, [C.c_uint64, C.c_size_t, C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]),
    "mh_msm_dev": (C.c_int, [C.c_uint64, C.c_size_t, C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]),
    "mh_msm_batch_dev": (C.c_int, [C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "mh_msm_batch": (C.c_int, [C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "mh_msm_batch_sharded_dev": (C.c_int, [C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "mh_g1_to_affine": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]),
    "mh_g1_sum": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p]),
    "mh_marlin_index": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, _u64p]),
    "mh_marlin_index_pc": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_int, _u64p]),
    "mh_marlin_pk_free": (C.c_int, [C.c_uint64]),
    "mh_marlin_pk_info": (C.c_int, [C.c_uint64, _u64p]),
    "mh_marlin_vk_bytes": (C.c_int, [C.c_uint64, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "mh_marlin_prove": (C.c_int, [C.c_uint64, C.c_void_p, C.c_void_p, C.c_char_p, C.c_int, C.c_void_p, C.c_size_t,
                                  C.POINTER(C.c_size_t)]),
    "mh_marlin_prove_dev": (C.c_int, [C.c_uint64, C.c_void_p, C.c_void_p, C.c_char_p, C.c_int, C.c_void_p, C.c_size_t,
                                      C.POINTER(C.c_size_t)]),
    "mh_marlin_prove_fs": (C.c_int, [C.c_uint64, C.c_void_p, C.c_void_p, C.c_char_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t,
                                     C.POINTER(C.c_size_t)]),
    "mh_marlin_verify_fs": (C.c_int, [C.c_char_p, C.c_size_t, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_void_p,
                                      C.POINTER(C.c_int)]),
    "mh_marlin_zk_draw_count": (C.c_int, [C.c_uint64, C.POINTER(C.c_size_t)]),
    "mh_marlin_prove_draws": (C.c_int, [C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                        C.POINTER(C.c_size_t)]),
    "mh_marlin_verify": (C.c_int, [C.c_char_p, C.c_size_t, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t, C.POINTER(C.c_int)]),
    "mh_pairing_product_is_one": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_int)]),
    "mh_marlin_proof_serialize": (C.c_int, [C.c_char_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "mh_marlin_proof_deserialize": (C.c_int, [C.c_char_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "mh_marlin_set_shard": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "mh_marlin_set_alltoall": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mh_marlin_set_alltoall_mode": (C.c_int, [C.c_int]),
    "mh_marlin_set_allgather_dev": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mh_marlin_probe_exchange_dev": (C.c_int, [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "mh_rccl_unique_id": (C.c_int, [C.c_void_p]),
    "mh_marlin_set_rccl": (C.c_int, [C.c_int, C.c_int, C.c_void_p]),
    "mh_marlin_rccl_sliced": (C.c_int, [C.c_int]),
    "mh_marlin_rccl_destroy": (C.c_int, []),
    "mh_marlin_rccl_info": (C.c_int, [_u64p, C.c_char_p, C.c_size_t]),
    "mh_marlin_exchange_stats": (C.c_int, [_u64p, C.POINTER(C.c_double), C.c_int]),
    "mh_ntt_dist_dev": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int]),
    "mh_msm_batch_sliced_dev": (C.c_int, [C.c_uint64, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "mh_marlin_probe_allgather": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p]),
    "mh_marlin_get_poly": (C.c_int, [C.c_uint64, C.c_char_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "mh_g2_bases_upload": (C.c_int, [C.c_int, C.c_void_p, C.c_size_t, _u64p]),
    "mh_g2_srs_powers": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, _u64p]),
    "mh_g2_bases_download": (C.c_int, [C.c_uint64, C.c_size_t, C.c_size_t, C.c_void_p]),
    "mh_g2_bases_free": (C.c_int, [C.c_uint64]),
    "mh_g2_msm": (C.c_int, [C.c_uint64, C.c_size_t, C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.POINTER(C.c_int)]),
    "mh_prof_enable": (C.c_int, [C.c_int]),
    "mh_prof_reset": (C.c_int, []),
    "mh_prof_get": (C.c_int, [C.c_int, C.POINTER(C.c_double), _u64p]),
    "mh_check_level": (C.c_int, [C.c_int]),
    "mh_check_report": (C.c_int, [C.c_char_p, C.c_size_t, _u64p]),
}
# include/marlin_hip_testhooks.h: exported by libmarlin_hip[_bn254]_testhooks.so only (fault injection, device self-test); bound
# when the loaded library has them -- a test that needs a hook runs with MARLIN_AMD_LIB pointing at the hooks library
HOOK_SYMBOLS = {
    "mh_selftest_fq30": (C.c_int, [C.c_uint64, C.c_uint64, _u64p]),
    "mh_debug_fail_scratch": (C.c_int, [C.c_int, _u64p]),
    "mh_debug_poison_scratch": (C.c_int, [C.c_int]),
    "mh_debug_corrupt": (C.c_int, [C.c_int]),
}
HOOKS_LIB_PATH = os.path.join(_HERE, {"bls12_381": "libmarlin_hip_testhooks.so", "bn254": "libmarlin_hip_bn254_testhooks.so"}[CURVE])


class MarlinHipError(RuntimeError):
    pass


_lib = None


def load():
    """dlopen libmarlin_hip.so and bind every declared symbol; raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MarlinHipError(
            "libmarlin_hip.so not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C marlin_amd/csrc`. There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    for name, (res, args) in HOOK_SYMBOLS.items():
        fn = getattr(lib, name, None)
        if fn is not None:
            fn.restype = res
            fn.argtypes = args
    _lib = lib
    cid, frl, fql, adic = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    lib.mh_curve_info(C.byref(cid), C.byref(frl), C.byref(fql), C.byref(adic))
    global CURVE_ID, FQ_LIMBS, FR_TWO_ADICITY
    CURVE_ID, FQ_LIMBS, FR_TWO_ADICITY = cid.value, fql.value, adic.value
    return lib


CURVE_ID, FQ_LIMBS, FR_TWO_ADICITY = 0, 6, 32


def check(rc, what=""):
    if rc != 0:
        msg = load().mh_last_error()
        raise MarlinHipError("%s failed (code %d): %s" % (what or "marlin_hip call", rc, (msg or b"").decode()))
