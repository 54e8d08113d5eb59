"""Thin, typed wrappers over the C ABI (no arithmetic happens in Python)."""
import ctypes as C
import numpy as np
from . import _lib

_R_MOD = {"bls12_381": 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001,
          "bn254": 21888242871839275222246405745257275088548364400416034343698204186575808495617}[_lib.CURVE]
# 2^256 mod r: the Montgomery representation of 1 in Fr
FR_ONE_MONT = np.array([(((1 << 256) % _R_MOD) >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)


def _curve_id():
    _L()
    return _lib.CURVE_ID


def _fql():
    _L()
    return _lib.FQ_LIMBS


def _L():
    return _lib.load()


def init(device=0):
    _lib.check(_L().mh_init(int(device)), "mh_init")


def shutdown():
    _lib.check(_L().mh_shutdown(), "mh_shutdown")


def synchronize():
    _lib.check(_L().mh_synchronize(), "mh_synchronize")


def device_info():
    name = C.create_string_buffer(256)
    cu = C.c_int()
    mem = C.c_size_t()
    _lib.check(_L().mh_device_info(name, 256, C.byref(cu), C.byref(mem)), "mh_device_info")
    return {"name": name.value.decode(), "cu_count": cu.value, "hbm_bytes": mem.value}


def _as_u64(a, limbs):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    if a.ndim != 2 or a.shape[1] != limbs:
        raise ValueError("expected uint64 array of shape (n, %d), got %r" % (limbs, a.shape))
    return a


class DeviceBuffer:
    """hipMalloc'd buffer owned by the library's device (mh_alloc / mh_free)."""

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        _lib.check(_L().mh_alloc(self.nbytes, C.byref(p)), "mh_alloc")
        self.ptr = p.value or 0

    @classmethod
    def from_numpy(cls, arr):
        arr = np.ascontiguousarray(arr)
        b = cls(arr.nbytes)
        b.upload(arr)
        return b

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        _lib.check(_L().mh_memcpy_h2d(self.ptr, arr.ctypes.data, arr.nbytes), "mh_memcpy_h2d")

    def download(self, shape, dtype=np.uint64):
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes
        _lib.check(_L().mh_memcpy_d2h(out.ctypes.data, self.ptr, out.nbytes), "mh_memcpy_d2h")
        return out

    def free(self):
        if self.ptr:
            _lib.check(_L().mh_free(self.ptr), "mh_free")
            self.ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _log2_exact(n):
    l = int(n).bit_length() - 1
    if n <= 0 or (1 << l) != n:
        raise ValueError("length must be a power of two, got %d" % n)
    return l


def ntt(evals_or_coeffs, inverse=False):
    """Host-buffer NTT (GeneralEvaluationDomain::fft / ifft): (n,4) uint64 Montgomery Fr."""
    a = _as_u64(evals_or_coeffs, 4).copy()
    log_n = _log2_exact(a.shape[0])
    _lib.check(_L().mh_ntt(_curve_id(), a.ctypes.data, log_n, 1 if inverse else 0), "mh_ntt")
    return a


def intt(evals):
    return ntt(evals, inverse=True)


def coset_ntt(coeffs_or_evals, inverse=False):
    """Radix2EvaluationDomain::coset_fft (inverse=False) / coset_ifft (inverse=True) of an (n,4) uint64 Montgomery array."""
    a = _as_u64(coeffs_or_evals, 4).copy()
    log_n = _log2_exact(a.shape[0])
    _lib.check(_L().mh_ntt_coset(_curve_id(), a.ctypes.data, log_n, 1 if inverse else 0), "mh_ntt_coset")
    return a


def ntt_dev(d_in, d_out, log_n, inverse=False):
    """Device-resident NTT; d_in / d_out are DeviceBuffer or raw device pointers."""
    pi = d_in.ptr if isinstance(d_in, DeviceBuffer) else int(d_in)
    po = d_out.ptr if isinstance(d_out, DeviceBuffer) else int(d_out)
    _lib.check(_L().mh_ntt_dev(_curve_id(), pi, po, int(log_n), 1 if inverse else 0), "mh_ntt_dev")


class Bases:
    """An SRS slice resident on the device (KZG10 powers_of_g): mh_bases_upload."""

    def __init__(self, xy_mont):
        a = _as_u64(xy_mont, 2 * _fql())
        h = C.c_uint64()
        _lib.check(_L().mh_bases_upload(_curve_id(), a.ctypes.data, a.shape[0], C.byref(h)), "mh_bases_upload")
        self.handle = h.value
        self.n = a.shape[0]

    @classmethod
    def from_serialized(cls, data, n, compressed=True):
        """n points in ark-serialize 0.3's image (`Vec<G1Affine>` without its length prefix; compressed: x with the sign /
        infinity flags, uncompressed: x || y), decoded and validated on the device (mh_bases_upload_serialized)."""
        data = bytes(data)
        item = (1 if compressed else 2) * 8 * _fql()
        assert len(data) == n * item, "expected %d bytes, got %d" % (n * item, len(data))
        h = C.c_uint64()
        _lib.check(_L().mh_bases_upload_serialized(_curve_id(), data, n, 1 if compressed else 0, C.byref(h)), "mh_bases_upload_serialized")
        self = cls.__new__(cls)
        self.handle, self.n = h.value, int(n)
        return self

    @classmethod
    def srs_powers(cls, tau_mont, n, scale_mont=None, first=0):
        """[scale * tau^(first+i)]G for i < n, generated on the device (KZG10::setup's powers_of_g).
        tau_mont / scale_mont: (4,) uint64 Montgomery Fr; scale defaults to one."""
        tau = np.ascontiguousarray(tau_mont, dtype=np.uint64).reshape(4)
        if scale_mont is None:
            scale_mont = FR_ONE_MONT
        sc = np.ascontiguousarray(scale_mont, dtype=np.uint64).reshape(4)
        h = C.c_uint64()
        _lib.check(_L().mh_srs_powers(_curve_id(), tau.ctypes.data, sc.ctypes.data, int(first), int(n), C.byref(h)), "mh_srs_powers")
        self = cls.__new__(cls)
        self.handle = h.value
        self.n = int(n)
        return self

    def precompute(self, window_bits=0):
        """Build the fixed-base window table (mh_bases_precompute); later MSMs against this set use it."""
        _lib.check(_L().mh_bases_precompute(self.handle, int(window_bits)), "mh_bases_precompute")
        return self

    def table_info(self):
        """(window_bits, windows, table_bytes) of the fixed-base table, zeros if none (mh_bases_table_info)."""
        c, w, b = C.c_uint32(), C.c_uint32(), C.c_uint64()
        _lib.check(_L().mh_bases_table_info(self.handle, C.byref(c), C.byref(w), C.byref(b)), "mh_bases_table_info")
        return c.value, w.value, b.value

    def download(self, offset=0, n=None):
        n = self.n - offset if n is None else n
        out = np.zeros((n, 2 * _fql()), dtype=np.uint64)
        _lib.check(_L().mh_bases_download(self.handle, int(offset), int(n), out.ctypes.data), "mh_bases_download")
        return out

    def free(self):
        if self.handle:
            _lib.check(_L().mh_bases_free(self.handle), "mh_bases_free")
            self.handle = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class G2Bases:
    """G2 points resident on the device (the verifier side of the SRS): mh_g2_bases_upload.  A point is
    x.c0 || x.c1 || y.c0 || y.c1, Montgomery Fq limbs, (n, 4 * FQ_LIMBS) uint64."""

    def __init__(self, xy_mont):
        a = _as_u64(xy_mont, 4 * _fql())
        h = C.c_uint64()
        _lib.check(_L().mh_g2_bases_upload(_curve_id(), a.ctypes.data, a.shape[0], C.byref(h)), "mh_g2_bases_upload")
        self.handle, self.n = h.value, a.shape[0]

    @classmethod
    def srs_powers(cls, gen_xy_mont, tau_mont, n, scale_mont=None, first=0):
        """[scale * tau^(first+i)]H for i < n (KZG10::setup's powers_of_h / neg_powers_of_h for a known tau)."""
        g = np.ascontiguousarray(gen_xy_mont, dtype=np.uint64).reshape(4 * _fql())
        tau = np.ascontiguousarray(tau_mont, dtype=np.uint64).reshape(4)
        sc = None if scale_mont is None else np.ascontiguousarray(scale_mont, dtype=np.uint64).reshape(4)
        h = C.c_uint64()
        _lib.check(_L().mh_g2_srs_powers(_curve_id(), g.ctypes.data, tau.ctypes.data, None if sc is None else sc.ctypes.data,
                                         int(first), int(n), C.byref(h)), "mh_g2_srs_powers")
        self = cls.__new__(cls)
        self.handle, self.n = h.value, int(n)
        return self

    def download(self, offset=0, n=None):
        n = self.n - offset if n is None else n
        out = np.zeros((n, 4 * _fql()), dtype=np.uint64)
        _lib.check(_L().mh_g2_bases_download(self.handle, int(offset), int(n), out.ctypes.data), "mh_g2_bases_download")
        return out

    def free(self):
        if self.handle:
            _lib.check(_L().mh_g2_bases_free(self.handle), "mh_g2_bases_free")
            self.handle = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def g2_msm(bases, scalars, base_offset=0, montgomery=True):
    """VariableBaseMSM::multi_scalar_mul over G2: (affine x.c0||x.c1||y.c0||y.c1 as (4*FQ_LIMBS,) uint64, is_infinity)."""
    s = _as_u64(scalars, 4)
    out = np.zeros(4 * _fql(), dtype=np.uint64)
    inf = C.c_int(0)
    _lib.check(_L().mh_g2_msm(bases.handle, int(base_offset), s.ctypes.data, 1 if montgomery else 0, s.shape[0],
                              out.ctypes.data, C.byref(inf)), "mh_g2_msm")
    return out, bool(inf.value)


def msm(bases, scalars, base_offset=0, montgomery=True):
    """VariableBaseMSM::multi_scalar_mul: returns Jacobian X||Y||Z as (18,) uint64 (Montgomery)."""
    s = _as_u64(scalars, 4)
    out = np.zeros(3 * _fql(), dtype=np.uint64)
    _lib.check(_L().mh_msm(bases.handle, int(base_offset), s.ctypes.data, 1 if montgomery else 0, s.shape[0],
                           out.ctypes.data), "mh_msm")
    return out


def msm_dev(bases, d_scalars, n, base_offset=0, montgomery=True):
    p = d_scalars.ptr if isinstance(d_scalars, DeviceBuffer) else int(d_scalars)
    out = np.zeros(3 * _fql(), dtype=np.uint64)
    _lib.check(_L().mh_msm_dev(bases.handle, int(base_offset), p, 1 if montgomery else 0, int(n), out.ctypes.data),
               "mh_msm_dev")
    return out


def msm_batch_dev(jobs, montgomery=True):
    """jobs: [(Bases, base_offset, DeviceBuffer or device pointer, n)] -> (len(jobs), 18) uint64 Jacobian results,
    computed by one batched launch sequence (mh_msm_batch_dev)."""
    k = len(jobs)
    handles = (C.c_uint64 * k)(*[j[0].handle for j in jobs])
    offs = (C.c_size_t * k)(*[int(j[1]) for j in jobs])
    ptrs = (C.c_void_p * k)(*[(j[2].ptr if isinstance(j[2], DeviceBuffer) else int(j[2])) for j in jobs])
    ns = (C.c_size_t * k)(*[int(j[3]) for j in jobs])
    out = np.zeros((k, 3 * _fql()), dtype=np.uint64)
    _lib.check(_L().mh_msm_batch_dev(k, handles, offs, ptrs, ns, 1 if montgomery else 0, out.ctypes.data), "mh_msm_batch_dev")
    return out


def msm_batch_sharded_dev(jobs, montgomery=True):
    """msm_batch_dev with every job sharded by bucket range over the ranks registered with mh_marlin_set_shard
    (marlin_amd.dist.enable_sharded_prove); every rank passes the same jobs and receives the combined results."""
    k = len(jobs)
    handles = (C.c_uint64 * k)(*[j[0].handle for j in jobs])
    offs = (C.c_size_t * k)(*[int(j[1]) for j in jobs])
    ptrs = (C.c_void_p * k)(*[(j[2].ptr if isinstance(j[2], DeviceBuffer) else int(j[2])) for j in jobs])
    ns = (C.c_size_t * k)(*[int(j[3]) for j in jobs])
    out = np.zeros((k, 3 * _fql()), dtype=np.uint64)
    _lib.check(_L().mh_msm_batch_sharded_dev(k, handles, offs, ptrs, ns, 1 if montgomery else 0, out.ctypes.data), "mh_msm_batch_sharded_dev")
    return out


def msm_path_counts():
    """(fixed-base groups, variable-base groups) served since mh_init (mh_msm_path_counts)."""
    a, b = C.c_uint64(), C.c_uint64()
    _lib.check(_L().mh_msm_path_counts(C.byref(a), C.byref(b)), "mh_msm_path_counts")
    return a.value, b.value


def g1_to_affine(xyz):
    """GroupProjective::into_affine on the host: returns ((12,) uint64 x||y Montgomery, is_infinity)."""
    xyz = np.ascontiguousarray(xyz, dtype=np.uint64)
    out = np.zeros(2 * _fql(), dtype=np.uint64)
    inf = C.c_int()
    _lib.check(_L().mh_g1_to_affine(xyz.ctypes.data, out.ctypes.data, C.byref(inf)), "mh_g1_to_affine")
    return out, bool(inf.value)


def prof_enable(on=True, families=None):
    """HIP-event timing per kernel family (mh_prof_enable); families = the ones to record (None = all)."""
    v = 0 if not on else (1 if families is None else sum(1 << (int(f) + 1) for f in families))
    _lib.check(_L().mh_prof_enable(v), "mh_prof_enable")


def prof_reset():
    _lib.check(_L().mh_prof_reset(), "mh_prof_reset")


def prof_get(family):
    ms = C.c_double()
    n = C.c_uint64()
    _lib.check(_L().mh_prof_get(int(family), C.byref(ms), C.byref(n)), "mh_prof_get")
    return ms.value, n.value
