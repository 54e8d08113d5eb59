"""Python mirror of the reference's `Marlin<F, PC, FS>` surface (src/lib.rs:64-311) over the C ABI:
universal_setup / index / prove with PC = MarlinKZG10<Bls12_381>, FS = SimpleHashFiatShamirRng<
Blake2s, ChaChaRng>.  Plumbing only: arguments are marshalled into the arkworks in-memory layout
and handed to libmarlin_hip.so; nothing is computed here."""
import ctypes as C
import numpy as np
from . import _lib
from .api import Bases, FR_ONE_MONT

from .api import _R_MOD as R_MOD
_MONT_R = (1 << 256) % R_MOD


def fr_mont(x):
    """canonical int -> (4,) uint64 Montgomery limbs."""
    v = (x % R_MOD) * _MONT_R % R_MOD
    return np.array([(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)


class _R1csMatrices(C.Structure):
    _fields_ = [("num_constraints", C.c_uint64), ("num_instance", C.c_uint64),
                ("row_ptr", C.c_void_p * 3), ("col", C.c_void_p * 3), ("val", C.c_void_p * 3)]


def max_degree(num_constraints, num_variables, num_non_zero):
    """AHPForR1CS::max_degree (src/ahp/mod.rs:71-93)."""
    def np2(n):
        s = 1
        while s < n:
            s *= 2
        return s
    h = np2(max(num_variables, num_constraints))
    k = np2(num_non_zero)
    return max(2 * h - 1, 3 * h - 1, h, k - 1)


class UniversalSRS:
    """KZG10 universal parameters for a known (test) tau: powers_of_g and powers_of_gamma_g on the device."""

    def __init__(self, max_deg, tau, gamma, full_gamma=False):
        self.max_degree = int(max_deg)
        self.tau, self.gamma = int(tau) % R_MOD, int(gamma) % R_MOD
        self.powers_of_g = Bases.srs_powers(fr_mont(self.tau), self.max_degree + 1)
        # MarlinKZG10::trim keeps powers_of_gamma_g[0..3); SonicKZG10 also needs them at max_degree - bound + i
        self.powers_of_gamma_g = Bases.srs_powers(fr_mont(self.tau), self.max_degree + 2 if full_gamma else 3,
                                                  scale_mont=fr_mont(self.gamma))

    def verifier_key(self, pk, h_xy_mont, pc="marlin"):
        """The group elements of the trimmed verifier key (PC::trim under Marlin::index, src/lib.rs:101-148) for `verify`:
        g, gamma_g, h, beta_h = [tau]h (KZG10::setup's G2 side, on the device), and for the two enforced degree bounds
        |H| - 2 and |K| - 2 the shift powers -- MarlinKZG10: powers_of_g[max_degree - bound]; SonicKZG10:
        [tau^-(max_degree - bound)]h (neg_powers_of_h)."""
        from .api import G2Bases

        def g2_power(scale):
            b = G2Bases.srs_powers(h_xy_mont, fr_mont(self.tau), 1, scale_mont=fr_mont(scale))
            try:
                return b.download()[0]
            finally:
                b.free()
        shifts = [self.max_degree - (pk.H - 2), self.max_degree - (pk.K - 2)]
        if pc == "sonic":
            tinv = pow(self.tau, -1, R_MOD)
            sp = [g2_power(pow(tinv, d, R_MOD)) for d in shifts]
        else:
            sp = [self.powers_of_g.download(d, 1)[0] for d in shifts]
        return [self.powers_of_g.download(0, 1)[0], self.powers_of_gamma_g.download(0, 1)[0],
                np.ascontiguousarray(h_xy_mont, dtype=np.uint64).reshape(-1), g2_power(self.tau)] + sp


_Q_MOD = {"bls12_381": 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab,
          "bn254": 21888242871839275222246405745257275088696311157297823662689037894645226208583}[_lib.CURVE]
# the standard generators of G2 (BLS12-381: the one ark-bls12-381 0.3 and the IETF draft use; BN254: EIP-197's), x.c0, x.c1, y.c0, y.c1
_G2_GENERATOR = {
    "bls12_381": (0x024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8,
                  0x13e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e,
                  0x0ce5d527727d6e118cc9cdc6da2e351aadfd9baa8cbdd3a76d429a695160d12c923ac9cc3baca289e193548608b82801,
                  0x0606c4a02ea734cc32acd2b02bc28b99cb3e287e85a763af267492ab572e99ab3f370d275cec1da1aaa9075ff05f79be),
    "bn254": (10857046999023057135944570762232829481370756359578518086990519993285655852781,
              11559732032986387107991004021392285783925812861821192530917403151452391805634,
              8495653923123431417604973247489272438418190587263600148770280649306958101930,
              4082367875863433681332203403145435568316851327593401208105741076214120093531),
}[_lib.CURVE]


def g2_generator_mont():
    """The G2 generator as the (4 * FQ_LIMBS,) uint64 Montgomery limbs `UniversalSRS.verifier_key` and the mh_g2_* entry points
    take -- an `h` for a test / bench SRS (kzg10::setup draws h at random; any point of G2 serves)."""
    nl = 6 if _lib.CURVE == "bls12_381" else 4
    out = []
    for c in _G2_GENERATOR:
        v = c * (1 << (64 * nl)) % _Q_MOD
        out += [(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(nl)]
    return np.array(out, dtype=np.uint64)


def universal_setup(num_constraints, num_variables, num_non_zero, tau, gamma, pc="marlin"):
    """Marlin::universal_setup (src/lib.rs:79-96) with a caller-chosen tau (test/bench SRS)."""
    return UniversalSRS(max_degree(num_constraints, num_variables, num_non_zero), tau, gamma, full_gamma=(pc == "sonic"))


class IndexProverKey:
    def __init__(self, handle):
        self.handle = handle
        info = (C.c_uint64 * 8)()
        _lib.check(_lib.load().mh_marlin_pk_info(handle, info), "mh_marlin_pk_info")
        (self.H, self.K, self.X, self.num_non_zero, self.max_degree, self.srs_max_degree, self.num_constraints,
         self.num_instance) = [int(x) for x in info]

    def vk_bytes(self):
        n = C.c_size_t()
        _lib.check(_lib.load().mh_marlin_vk_bytes(self.handle, None, 0, C.byref(n)), "mh_marlin_vk_bytes")
        buf = (C.c_uint8 * n.value)()
        _lib.check(_lib.load().mh_marlin_vk_bytes(self.handle, buf, n.value, C.byref(n)), "mh_marlin_vk_bytes")
        return bytes(buf)

    def get_poly(self, label):
        """(len,4) uint64 Montgomery coefficients of a prover/indexer polynomial of the last proof."""
        n = C.c_size_t()
        _lib.check(_lib.load().mh_marlin_get_poly(self.handle, label.encode(), None, 0, C.byref(n)), "mh_marlin_get_poly")
        out = np.zeros((n.value, 4), dtype=np.uint64)
        _lib.check(_lib.load().mh_marlin_get_poly(self.handle, label.encode(), out.ctypes.data, n.value, C.byref(n)),
                   "mh_marlin_get_poly")
        return out

    def free(self):
        if self.handle:
            _lib.check(_lib.load().mh_marlin_pk_free(self.handle), "mh_marlin_pk_free")
            self.handle = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def index(srs, num_constraints, num_instance, matrices, pc="marlin"):
    """Marlin::index (src/lib.rs:100-148).  matrices = [(row_ptr uint64[nc+1], col uint32[nnz], val (nnz,4) uint64
    Montgomery or None for all-ones)] for A, B, C -- already padded and square."""
    keep = []
    m = _R1csMatrices()
    m.num_constraints = int(num_constraints)
    m.num_instance = int(num_instance)
    for k, (rp, col, val) in enumerate(matrices):
        rp = np.ascontiguousarray(rp, dtype=np.uint64)
        col = np.ascontiguousarray(col, dtype=np.uint32)
        keep += [rp, col]
        m.row_ptr[k] = rp.ctypes.data
        m.col[k] = col.ctypes.data if col.size else None
        if val is not None:
            val = np.ascontiguousarray(val, dtype=np.uint64)
            keep.append(val)
            m.val[k] = val.ctypes.data
        else:
            m.val[k] = None
    h = C.c_uint64()
    _lib.check(_lib.load().mh_marlin_index_pc(C.byref(m), srs.powers_of_g.handle, srs.powers_of_gamma_g.handle,
                                              {"marlin": 0, "sonic": 1}[pc], C.byref(h)), "mh_marlin_index_pc")
    pk = IndexProverKey(h.value)
    pk.srs = srs
    return pk


def proof_bytes_len(pc="marlin"):
    """length of the flat proof: G1 = 2 * Fq bytes + 1; Marlin commitment = 2 G1 + 1; Sonic commitment = 1 G1."""
    g1 = 2 * 8 * _lib.FQ_LIMBS + 1
    comm = (2 * g1 + 1) if pc == "marlin" else g1
    return 9 * comm + 4 * 32 + 2 * (g1 + 1 + 32)


PROOF_BYTES = 2143        # BLS12-381, MarlinKZG10


def prove(pk, instance_mont, witness_mont, zk_seed, zk_rounds=20):
    """Marlin::prove (src/lib.rs:151-311).  Returns the flat ToBytes-layout proof."""
    x = np.ascontiguousarray(instance_mont, dtype=np.uint64)
    w = np.ascontiguousarray(witness_mont, dtype=np.uint64)
    assert x.shape == (pk.num_instance, 4) and w.shape == (pk.num_constraints - pk.num_instance, 4), (x.shape, w.shape)
    out = (C.c_uint8 * 4096)()
    n = C.c_size_t()
    _lib.check(_lib.load().mh_marlin_prove(pk.handle, x.ctypes.data, w.ctypes.data, bytes(zk_seed), int(zk_rounds), out, 4096,
                                           C.byref(n)), "mh_marlin_prove")
    return bytes(out[:n.value])


class _FiatShamirC(C.Structure):
    _fields_ = [("user", C.c_void_p), ("initialize", C.c_void_p), ("absorb", C.c_void_p), ("next_u64", C.c_void_p)]


_FS_BYTES_T = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_uint8), C.c_size_t)
_FS_U64_T = C.CFUNCTYPE(C.c_uint64, C.c_void_p)


def _fs_struct(fs):
    """The caller's `FS: FiatShamirRng` (an object with initialize(bytes), absorb(bytes), next_u64() -> int) as mh_fiat_shamir.
    Returns (struct, keepalive, errors).  The C callbacks have no error channel (ctypes prints an exception raised inside one and
    returns 0 in its place): the FIRST exception of the caller's object is recorded in `errors` and every later callback becomes a
    no-op, so that prove_fs / verify_fs can re-raise it once the library call is back instead of handing out a proof (or a
    verdict) computed over a corrupted transcript."""
    errors = []

    def guard(f, default=None):
        def g(*a):
            if errors:
                return default
            try:
                return f(*a)
            except BaseException as e:          # noqa: BLE001 -- re-raised by the caller of the library
                errors.append(e)
                return default
        return g
    init = _FS_BYTES_T(guard(lambda _u, p, n: fs.initialize(bytes(p[:n]))))
    absorb = _FS_BYTES_T(guard(lambda _u, p, n: fs.absorb(bytes(p[:n]))))
    nxt = _FS_U64_T(guard(lambda _u: int(fs.next_u64()) & 0xffffffffffffffff, 0))
    st = _FiatShamirC(None, C.cast(init, C.c_void_p), C.cast(absorb, C.c_void_p), C.cast(nxt, C.c_void_p))
    return st, (init, absorb, nxt), errors


def prove_fs(pk, instance_mont, witness_mont, zk_seed, fs, zk_rounds=20):
    """Marlin<F, PC, FS>::prove for any `FS: FiatShamirRng` (src/lib.rs:64-70,151-155; mh_marlin_prove_fs): `fs` supplies
    initialize / absorb / next_u64."""
    x = np.ascontiguousarray(instance_mont, dtype=np.uint64)
    w = np.ascontiguousarray(witness_mont, dtype=np.uint64)
    assert x.shape == (pk.num_instance, 4) and w.shape == (pk.num_constraints - pk.num_instance, 4), (x.shape, w.shape)
    st, keep, errors = _fs_struct(fs)
    out = (C.c_uint8 * 4096)()
    n = C.c_size_t()
    rc = _lib.load().mh_marlin_prove_fs(pk.handle, x.ctypes.data, w.ctypes.data, bytes(zk_seed), int(zk_rounds), C.byref(st), out, 4096,
                                        C.byref(n))
    del keep
    if errors:
        raise errors[0]
    _lib.check(rc, "mh_marlin_prove_fs")
    return bytes(out[:n.value])


def zk_draw_count(pk):
    n = C.c_size_t()
    _lib.check(_lib.load().mh_marlin_zk_draw_count(pk.handle, C.byref(n)), "mh_marlin_zk_draw_count")
    return n.value


def prove_draws(pk, instance_mont, witness_mont, zk_draws_mont):
    """Marlin::prove with the caller's own `zk_rng` draws ((zk_draw_count(pk), 4) uint64 Montgomery, consumption order):
    the entry point for a host whose rng is not a ChaCha generator (src/lib.rs:151-155)."""
    x = np.ascontiguousarray(instance_mont, dtype=np.uint64)
    w = np.ascontiguousarray(witness_mont, dtype=np.uint64)
    d = np.ascontiguousarray(zk_draws_mont, dtype=np.uint64)
    assert x.shape == (pk.num_instance, 4) and w.shape == (pk.num_constraints - pk.num_instance, 4) and d.ndim == 2 and d.shape[1] == 4
    out = (C.c_uint8 * 4096)()
    n = C.c_size_t()
    _lib.check(_lib.load().mh_marlin_prove_draws(pk.handle, x.ctypes.data, w.ctypes.data, d.ctypes.data, d.shape[0], out, 4096,
                                                 C.byref(n)), "mh_marlin_prove_draws")
    return bytes(out[:n.value])


def prove_dev(pk, d_instance, d_witness, zk_seed, zk_rounds=20):
    """Marlin::prove with the formatted input and the witness already on the device (DeviceBuffer or device pointers):
    mh_marlin_prove_dev.  Returns the flat ToBytes-layout proof."""
    pi = d_instance.ptr if hasattr(d_instance, "ptr") else int(d_instance)
    pw = d_witness.ptr if hasattr(d_witness, "ptr") else int(d_witness)
    out = (C.c_uint8 * 4096)()
    n = C.c_size_t()
    _lib.check(_lib.load().mh_marlin_prove_dev(pk.handle, pi, pw, bytes(zk_seed), int(zk_rounds), out, 4096, C.byref(n)),
               "mh_marlin_prove_dev")
    return bytes(out[:n.value])


class _VerifierKeyC(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("g_xy", "gamma_g_xy", "h_xy", "beta_h_xy", "shift_power_h_xy", "shift_power_k_xy")]


def verify(vk_bytes, g_xy, gamma_g_xy, h_xy, beta_h_xy, shift_power_h_xy, shift_power_k_xy, public_input_mont, flat_proof, pc="marlin"):
    """Marlin::verify (src/lib.rs:315-433) on the host (mh_marlin_verify), MarlinKZG10 or SonicKZG10 on the library's curve:
    True = accept.  Group elements: affine Montgomery limbs as uint64 arrays (G1: 2 * FQ_LIMBS, G2: 4 * FQ_LIMBS); the two
    shift powers are G1 points for pc="marlin" and G2 points ([beta^-(max_degree - bound)]h) for pc="sonic";
    public_input_mont: (n, 4) uint64 (the unformatted input)."""
    arrs = [np.ascontiguousarray(a, dtype=np.uint64) for a in (g_xy, gamma_g_xy, h_xy, beta_h_xy, shift_power_h_xy, shift_power_k_xy)]
    vk = _VerifierKeyC(*[a.ctypes.data for a in arrs])
    pub = np.ascontiguousarray(public_input_mont, dtype=np.uint64).reshape(-1, 4)
    ok = C.c_int(0)
    _lib.check(_lib.load().mh_marlin_verify(bytes(vk_bytes), len(vk_bytes), C.byref(vk), {"marlin": 0, "sonic": 1}[pc], pub.ctypes.data,
                                            pub.shape[0], bytes(flat_proof), len(flat_proof), C.byref(ok)), "mh_marlin_verify")
    return bool(ok.value)


def verify_fs(vk_bytes, g_xy, gamma_g_xy, h_xy, beta_h_xy, shift_power_h_xy, shift_power_k_xy, public_input_mont, flat_proof, fs, pc="marlin"):
    """verify() with the caller's `FS: FiatShamirRng` (mh_marlin_verify_fs)."""
    arrs = [np.ascontiguousarray(a, dtype=np.uint64) for a in (g_xy, gamma_g_xy, h_xy, beta_h_xy, shift_power_h_xy, shift_power_k_xy)]
    vk = _VerifierKeyC(*[a.ctypes.data for a in arrs])
    pub = np.ascontiguousarray(public_input_mont, dtype=np.uint64).reshape(-1, 4)
    st, keep, errors = _fs_struct(fs)
    ok = C.c_int(0)
    rc = _lib.load().mh_marlin_verify_fs(bytes(vk_bytes), len(vk_bytes), C.byref(vk), {"marlin": 0, "sonic": 1}[pc], pub.ctypes.data,
                                         pub.shape[0], bytes(flat_proof), len(flat_proof), C.byref(st), C.byref(ok))
    del keep
    if errors:
        raise errors[0]
    _lib.check(rc, "mh_marlin_verify_fs")
    return bool(ok.value)


def pairing_product_is_one(g1_xy_mont, g2_xy_mont):
    """prod_i e(P_i, Q_i) == 1 on the host (mh_pairing_product_is_one); (n, 2 FQ_LIMBS) and (n, 4 FQ_LIMBS) uint64 arrays."""
    _lib.load()
    a = np.ascontiguousarray(g1_xy_mont, dtype=np.uint64).reshape(-1, 2 * _lib.FQ_LIMBS)
    b = np.ascontiguousarray(g2_xy_mont, dtype=np.uint64).reshape(-1, 4 * _lib.FQ_LIMBS)
    assert a.shape[0] == b.shape[0]
    ok = C.c_int(0)
    _lib.check(_lib.load().mh_pairing_product_is_one(a.ctypes.data, b.ctypes.data, a.shape[0], C.byref(ok)), "mh_pairing_product_is_one")
    return bool(ok.value)


def proof_serialize(flat_proof, pc="marlin"):
    """flat ToBytes proof (prove) -> ark-serialize `CanonicalSerialize for Proof` bytes (src/data_structures.rs:100-110)."""
    out = (C.c_uint8 * 4096)()
    n = C.c_size_t()
    _lib.check(_lib.load().mh_marlin_proof_serialize(bytes(flat_proof), len(flat_proof), {"marlin": 0, "sonic": 1}[pc], out, 4096,
                                                     C.byref(n)), "mh_marlin_proof_serialize")
    return bytes(out[:n.value])


def proof_deserialize(wire_bytes, pc="marlin"):
    """`CanonicalDeserialize for Proof` (validating) -> flat ToBytes proof."""
    out = (C.c_uint8 * 4096)()
    n = C.c_size_t()
    _lib.check(_lib.load().mh_marlin_proof_deserialize(bytes(wire_bytes), len(wire_bytes), {"marlin": 0, "sonic": 1}[pc], out, 4096,
                                                       C.byref(n)), "mh_marlin_proof_deserialize")
    return bytes(out[:n.value])


# ---- the reference's benchmark / test circuits as padded square R1CS (host-side input preparation) ----
def dummy_circuit(a, b, num_variables, num_constraints):
    """DummyCircuit of benches/bench.rs:26-66 after pad_input / make_matrices_square: returns
    (num_constraints_padded, num_instance, matrices, instance_mont, witness_mont)."""
    assert num_variables >= 3 and num_constraints >= 2
    ni = 2                                   # One, c
    nvars = ni + (2 + num_variables - 3)
    nc = max(num_constraints, nvars)
    extra_wit = nc - nvars if num_constraints >= nvars else 0
    n_rows_real = num_constraints - 1

    def mat(colidx):
        rp = np.zeros(nc + 1, dtype=np.uint64)
        rp[1:n_rows_real + 1] = np.arange(1, n_rows_real + 1, dtype=np.uint64)
        rp[n_rows_real + 1:] = n_rows_real
        return rp, np.full(n_rows_real, colidx, dtype=np.uint32), None
    matrices = [mat(ni + 0), mat(ni + 1), mat(1)]
    a, b = a % R_MOD, b % R_MOD
    inst = np.stack([FR_ONE_MONT, fr_mont(a * b)])
    am = fr_mont(a)
    wit = np.empty((nc - ni, 4), dtype=np.uint64)
    wit[0] = am
    wit[1] = fr_mont(b)
    wit[2:2 + num_variables - 3] = am
    wit[2 + num_variables - 3:] = FR_ONE_MONT        # make_matrices_square: dummy witnesses = one
    assert extra_wit == nc - ni - (2 + num_variables - 3)
    return nc, ni, matrices, inst, wit


def test_circuit(a, b, num_constraints, num_variables):
    """Circuit of src/test.rs:9-50 after padding: inputs (1, c, d, 0)."""
    ni = 4
    nvars = ni + (2 + num_variables - 3)
    nc = max(num_constraints, nvars)
    a, b = a % R_MOD, b % R_MOD
    c = a * b % R_MOD
    d = c * b % R_MOD
    col_a, col_b, col_c, col_d = ni, ni + 1, 1, 2

    def mat(main, last):
        rp = np.zeros(nc + 1, dtype=np.uint64)
        rp[1:num_constraints + 1] = np.arange(1, num_constraints + 1, dtype=np.uint64)
        rp[num_constraints + 1:] = num_constraints
        col = np.full(num_constraints, main, dtype=np.uint32)
        col[-1] = last
        return rp, col, None
    matrices = [mat(col_a, col_c), mat(col_b, col_b), mat(col_c, col_d)]
    inst = np.stack([FR_ONE_MONT, fr_mont(c), fr_mont(d), fr_mont(0)])
    wit = np.empty((nc - ni, 4), dtype=np.uint64)
    wit[:] = FR_ONE_MONT
    wit[0] = fr_mont(a)
    wit[1] = fr_mont(b)
    wit[2:2 + num_variables - 3] = fr_mont(a)
    return nc, ni, matrices, inst, wit
