"""ctypes wrapper of oracle/c/ref_hotpath.c (the C restatement).  Oracle only."""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libref_hotpath.so")
_lib = None


def build(force=False):
    src = os.path.join(_HERE, "c", "ref_hotpath.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", os.path.join(_HERE, "c")])
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        L.ref_ntt.restype = C.c_int
        L.ref_ntt.argtypes = [C.c_void_p, C.c_uint32, C.c_int]
        L.ref_ntt_mt.restype = C.c_int
        L.ref_ntt_mt.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_int]
        L.ref_msm.restype = C.c_int
        L.ref_msm.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_int, C.c_void_p]
        L.ref_g1_to_affine.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
        L.ref_g1_mul_gen.argtypes = [C.c_void_p, C.c_void_p]
        L.ref_bases_arith.restype = C.c_int
        L.ref_bases_arith.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.ref_fr_from_mont.argtypes = [C.c_void_p, C.c_size_t]
        L.ref_fr_to_mont.argtypes = [C.c_void_p, C.c_size_t]
        L.ref_fr_mul_vec.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        _lib = L
    return _lib


def _limbs(x, n):
    return np.array([(x >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(n)], dtype=np.uint64)


def ntt(arr, inverse=False, threads=1):
    """(n,4) uint64 Montgomery Fr -> transformed copy (threads > 1: the stages' butterflies split over OpenMP threads)."""
    a = np.ascontiguousarray(arr, dtype=np.uint64).copy()
    n = a.shape[0]
    log_n = n.bit_length() - 1
    assert 1 << log_n == n
    rc = lib().ref_ntt_mt(a.ctypes.data, log_n, 1 if inverse else 0, int(threads))
    assert rc == 0
    return a


def msm(bases_xy, scalars, montgomery=True, threads=1):
    b = np.ascontiguousarray(bases_xy, dtype=np.uint64)
    s = np.ascontiguousarray(scalars, dtype=np.uint64)
    n = min(b.shape[0], s.shape[0])
    out = np.zeros(18, dtype=np.uint64)
    rc = lib().ref_msm(b.ctypes.data, s.ctypes.data, 1 if montgomery else 0, n, threads, out.ctypes.data)
    assert rc == 0
    return out


def g1_to_affine(xyz):
    xyz = np.ascontiguousarray(xyz, dtype=np.uint64)
    out = np.zeros(12, dtype=np.uint64)
    inf = C.c_int()
    lib().ref_g1_to_affine(xyz.ctypes.data, out.ctypes.data, C.byref(inf))
    return out, bool(inf.value)


def g1_mul_gen(k):
    out = np.zeros(18, dtype=np.uint64)
    kk = _limbs(k, 4)
    lib().ref_g1_mul_gen(kk.ctypes.data, out.ctypes.data)
    return out


def bases_arith(n, a0=0x1234567, d=0xabcdef1):
    """P_i = [a0 + i d]G as (n,12) uint64 + the discrete logs (python ints)."""
    from .fields import R_MOD
    out = np.zeros((n, 12), dtype=np.uint64)
    la, ld = _limbs(a0, 4), _limbs(d, 4)   # keep alive across the call
    rc = lib().ref_bases_arith(la.ctypes.data, ld.ctypes.data, n, out.ctypes.data)
    assert rc == 0
    return out, [(a0 + i * d) % R_MOD for i in range(n)]
