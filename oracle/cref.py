"""ctypes wrapper of oracle/c/ref_hotpath.c (the C restatement).  Oracle only."""
import ctypes as C
import os
import subprocess
import numpy as np

from . import fields as _F

_HERE = os.path.dirname(os.path.abspath(__file__))
# one library per curve (the oracle picks its curve at import: ORACLE_CURVE)
_SO = os.path.join(_HERE, "_build", {"bls12_381": "libref_hotpath.so", "bn254": "libref_hotpath_bn254.so"}[_F.CURVE])
_lib = None
FQL = _F.FQ_LIMBS64


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
        L.ref_fr_lincomb.restype = C.c_int
        L.ref_fr_lincomb.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.ref_fr_div_linear.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        L.ref_fr_eval.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        L.ref_fr_add_at.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.ref_srs_powers.restype = C.c_int
        L.ref_srs_powers.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        L.ref_curve_id.restype = C.c_int
        assert L.ref_curve_id() == {"bls12_381": 0, "bn254": 1}[_F.CURVE]
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
    out = np.zeros(3 * FQL, dtype=np.uint64)
    rc = lib().ref_msm(b.ctypes.data, s.ctypes.data, 1 if montgomery else 0, n, threads, out.ctypes.data)
    assert rc == 0
    return out


def g1_to_affine(xyz):
    xyz = np.ascontiguousarray(xyz, dtype=np.uint64)
    out = np.zeros(2 * FQL, dtype=np.uint64)
    inf = C.c_int()
    lib().ref_g1_to_affine(xyz.ctypes.data, out.ctypes.data, C.byref(inf))
    return out, bool(inf.value)


def g1_mul_gen(k):
    out = np.zeros(3 * FQL, dtype=np.uint64)
    kk = _limbs(k, 4)
    lib().ref_g1_mul_gen(kk.ctypes.data, out.ctypes.data)
    return out


def bases_arith(n, a0=0x1234567, d=0xabcdef1):
    """P_i = [a0 + i d]G as (n, 2 * FQL) uint64 + the discrete logs (python ints)."""
    from .fields import R_MOD
    out = np.zeros((n, 2 * FQL), dtype=np.uint64)
    la, ld = _limbs(a0, 4), _limbs(d, 4)   # keep alive across the call
    rc = lib().ref_bases_arith(la.ctypes.data, ld.ctypes.data, n, out.ctypes.data)
    assert rc == 0
    return out, [(a0 + i * d) % R_MOD for i in range(n)]


def _fr1(x_canonical):
    """canonical int -> (4,) uint64 Montgomery"""
    return _limbs(_F.fr_to_mont(x_canonical % _F.R_MOD), 4)


def _fr1_int(row):
    return _F.fr_from_mont(sum(int(row[k]) << (64 * k) for k in range(4)))


def lincomb(terms, n=None, threads=1):
    """terms: [(coef canonical int, (len,4) uint64 Montgomery coefficients)] -> (n,4) sum_t coef_t * poly_t (n defaults to
    the longest term)."""
    terms = [(c, np.ascontiguousarray(a, dtype=np.uint64)) for c, a in terms]
    if n is None:
        n = max([len(a) for _, a in terms] + [0])
    out = np.zeros((n, 4), dtype=np.uint64)
    if n == 0 or not terms:
        return out
    ptrs = (C.c_void_p * len(terms))(*[a.ctypes.data for _, a in terms])
    lens = (C.c_size_t * len(terms))(*[min(len(a), n) for _, a in terms])
    coef = np.stack([_fr1(c) for c, _ in terms])
    rc = lib().ref_fr_lincomb(out.ctypes.data, n, len(terms), ptrs, lens, coef.ctypes.data, int(threads))
    assert rc == 0
    return out


def div_linear(p, z):
    """(len,4) Montgomery coefficients, z canonical int -> (quotient (len-1,4), p(z) canonical int)"""
    p = np.ascontiguousarray(p, dtype=np.uint64)
    q = np.zeros((max(len(p) - 1, 0), 4), dtype=np.uint64)
    rem = np.zeros(4, dtype=np.uint64)
    zz = _fr1(z)
    if len(p):
        lib().ref_fr_div_linear(q.ctypes.data, p.ctypes.data, len(p), zz.ctypes.data, rem.ctypes.data)
    return q, _fr1_int(rem)


def poly_eval(p, z):
    p = np.ascontiguousarray(p, dtype=np.uint64)
    out = np.zeros(4, dtype=np.uint64)
    zz = _fr1(z)
    if len(p):
        lib().ref_fr_eval(p.ctypes.data, len(p), zz.ctypes.data, out.ctypes.data)
    return _fr1_int(out)


def add_at(dst, off, src):
    """dst[off : off + len(src)] += src, in place ((n,4) uint64 Montgomery)"""
    src = np.ascontiguousarray(src, dtype=np.uint64)
    assert dst.flags["C_CONTIGUOUS"] and off + len(src) <= len(dst)
    if len(src):
        lib().ref_fr_add_at(dst.ctypes.data, off, src.ctypes.data, len(src))


def srs_powers(tau, n, scale=1, threads=1):
    """[scale * tau^i]G, i < n, as (n, 2 * FQL) uint64 affine Montgomery (KZG10::setup's powers for a known tau)."""
    out = np.zeros((n, 2 * FQL), dtype=np.uint64)
    t, s = _limbs(tau % _F.R_MOD, 4), _limbs(scale % _F.R_MOD, 4)
    rc = lib().ref_srs_powers(t.ctypes.data, s.ctypes.data, n, int(threads), out.ctypes.data)
    assert rc == 0, rc
    return out
