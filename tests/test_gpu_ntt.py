"""Parity of the HIP NTT (through the C ABI) against the oracle's radix-2 / naive DFT.
Bit-exact (integer work)."""
import numpy as np
import pytest
from oracle import fields as F
from oracle import poly as OP
from tests.util import fr_to_np, np_to_fr, rand_fr

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("log_n", list(range(0, 13)))
def test_ntt_matches_oracle(gpu, log_n):
    n = 1 << log_n
    v = rand_fr(n, 100 + log_n)
    got = np_to_fr(gpu.ntt(fr_to_np(v)))
    assert got == OP.ntt(v, log_n)
    goti = np_to_fr(gpu.intt(fr_to_np(v)))
    assert goti == OP.ntt(v, log_n, inverse=True)


@pytest.mark.parametrize("log_n", [1, 3, 6])
def test_ntt_matches_naive_dft(gpu, log_n):
    v = rand_fr(1 << log_n, 7)
    assert np_to_fr(gpu.ntt(fr_to_np(v))) == OP.dft_naive(v, log_n)


def test_ntt_edge_values(gpu):
    r = F.R_MOD
    for v in ([0] * 16, [1] * 16, [r - 1] * 16, [F.FR_MONT_R % r, F.FR_MONT_R2 % r, 0, 1] * 4,
              [1] + [0] * 15):
        assert np_to_fr(gpu.ntt(fr_to_np(v))) == OP.ntt(v, 4)


@pytest.mark.parametrize("log_n", [14, 17, 20, 22])
def test_ntt_roundtrip_and_spot_large(gpu, log_n):
    """size-independent properties at sizes the Python oracle cannot transform:
    ifft(fft(x)) == x, and spot evaluations p(w^i) by Horner for a sparse p."""
    n = 1 << log_n
    rng = np.random.default_rng(log_n)
    # random Montgomery limbs < r: draw 62-bit top limb (always < r's top limb 0x73ed...)
    x = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
    x[:, 3] &= np.uint64((1 << 61) - 1)
    y = gpu.ntt(x)
    back = gpu.intt(y)
    assert np.array_equal(back, x)
    # sparse polynomial: few nonzero coefficients -> evaluate exactly with Python ints
    coeffs = {0: 5, 1: 7, n // 2 + 3: 11, n - 1: 13}
    dense = np.zeros((n, 4), dtype=np.uint64)
    for k, c in coeffs.items():
        dense[k] = fr_to_np([c])[0]
    ev = gpu.ntt(dense)
    w = F.root_of_unity(log_n)
    for i in [0, 1, 2, n // 3, n - 1]:
        wi = pow(w, i, F.R_MOD)
        want = sum(c * pow(wi, k, F.R_MOD) for k, c in coeffs.items()) % F.R_MOD
        assert np_to_fr(ev[i:i + 1])[0] == want


def test_ntt_linearity(gpu):
    log_n = 15
    n = 1 << log_n
    a = rand_fr(n, 1)
    b = rand_fr(n, 2)
    s = [(x + y) % F.R_MOD for x, y in zip(a, b)]
    fa = np_to_fr(gpu.ntt(fr_to_np(a)))
    fb = np_to_fr(gpu.ntt(fr_to_np(b)))
    fs = np_to_fr(gpu.ntt(fr_to_np(s)))
    assert fs == [(x + y) % F.R_MOD for x, y in zip(fa, fb)]


def test_ntt_dev_out_of_place(gpu):
    log_n = 18
    n = 1 << log_n
    v = fr_to_np(rand_fr(n, 9))
    din = gpu.DeviceBuffer.from_numpy(v)
    dout = gpu.DeviceBuffer(v.nbytes)
    gpu.ntt_dev(din, dout, log_n)
    assert np.array_equal(din.download(v.shape), v)          # input untouched
    assert np.array_equal(dout.download(v.shape), gpu.ntt(v))
    gpu.ntt_dev(dout, dout, log_n, inverse=True)               # in place
    assert np.array_equal(dout.download(v.shape), v)


@pytest.mark.parametrize("log_n", [0, 1, 4, 9, 12])
def test_coset_ntt_matches_oracle(gpu, log_n):
    """mh_ntt_coset = Radix2EvaluationDomain::{coset_fft, coset_ifft} (multiplicative generator 7 / 5): against the oracle,
    against the definition p(g w^i) by Horner, and as a round trip."""
    n = 1 << log_n
    dom = OP.Domain(n)
    v = rand_fr(n, 300 + log_n)
    got = np_to_fr(gpu.coset_ntt(fr_to_np(v)))
    assert got == dom.coset_fft(v)
    g = F.FR_GENERATOR
    for i in {0, n // 2, n - 1}:
        assert got[i] == OP.poly_eval(v, g * dom.element(i) % F.R_MOD)
    assert np_to_fr(gpu.coset_ntt(fr_to_np(v), inverse=True)) == dom.coset_ifft(v)
    assert np_to_fr(gpu.coset_ntt(fr_to_np(got), inverse=True)) == v


def test_coset_ntt_large_round_trip_and_spot(gpu):
    log_n = 20
    n = 1 << log_n
    rng = np.random.default_rng(5)
    x = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
    x[:, 3] &= np.uint64((1 << 61) - 1)
    y = gpu.coset_ntt(x)
    assert np.array_equal(gpu.coset_ntt(y, inverse=True), x)
    # a sparse polynomial: evaluations on the coset by the definition
    sp = np.zeros((n, 4), dtype=np.uint64)
    idx = [0, 1, 12345, n - 1]
    coeffs = [3, 5, 7, 11]
    sp[idx] = fr_to_np(coeffs)
    ev = np_to_fr(gpu.coset_ntt(sp)[[0, 1, 777, n - 1]])
    w = F.root_of_unity(log_n)
    for e, i in zip(ev, [0, 1, 777, n - 1]):
        pt = F.FR_GENERATOR * pow(w, i, F.R_MOD) % F.R_MOD
        assert e == sum(c * pow(pt, k, F.R_MOD) for c, k in zip(coeffs, idx)) % F.R_MOD


@pytest.mark.parametrize("log_n,in_len", [(10, 1), (10, 0), (12, 1025), (12, 4096), (14, 4097), (16, 3 * (1 << 14) + 1)])
def test_ntt_len_equals_the_padded_transform(gpu, log_n, in_len):
    """mh_ntt_len (the seam for ark-poly's fft_in_place, which first zero-pads the caller's Vec to the domain): uploading only
    the in_len coefficients gives the transform of the padded vector, forward and inverse; what lies beyond in_len in the
    caller's buffer is not read."""
    import ctypes as C
    from marlin_amd import _lib
    lib = _lib.load()
    n = 1 << log_n
    rng = np.random.default_rng(log_n * 7919 + in_len)
    x = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
    x[:, 3] &= np.uint64((1 << 59) - 1)
    padded = x.copy()
    padded[in_len:] = 0
    for inverse in (0, 1):
        want = gpu.ntt(padded) if not inverse else gpu.intt(padded)
        buf = x.copy()                      # garbage beyond in_len: must not matter
        _lib.check(lib.mh_ntt_len(_lib.CURVE_ID, buf.ctypes.data, in_len, log_n, inverse), "mh_ntt_len")
        assert np.array_equal(buf, want)
    assert lib.mh_ntt_len(_lib.CURVE_ID, x.ctypes.data, n + 1, log_n, 0) != 0          # in_len beyond the domain is refused


def _host_threads():
    import os
    try:
        return max(1, min(16, len(os.sched_getaffinity(0))))
    except AttributeError:
        return max(1, min(16, os.cpu_count() or 1))


@pytest.mark.parametrize("log_n", [20, 23])
def test_ntt_large_matches_the_c_restatement(gpu, log_n):
    """the transform kernels themselves, at the sizes of a 2^20-constraint proof -- 2^20 (|H|: one-stage rounds,
    Shoup twiddle products) and 2^23 (2|K|: two-stage rounds, Montgomery products) -- forward and inverse, element for element
    against the C restatement of the radix-2 transform (oracle/c/ref_hotpath.c, itself pinned to the Python oracle and the naive DFT
    by tests/test_oracle_c.py).  What `GeneralEvaluationDomain::{fft, ifft}` return at /root/reference src/ahp/prover.rs:532-535,685.
    Until round 5 these sizes were pinned only through whole proofs (a mismatch said "proof differs", not "this kernel differs")."""
    from oracle import cref
    n = 1 << log_n
    rng = np.random.default_rng(1000 + log_n)
    x = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
    x[:, 3] &= np.uint64((1 << 61) - 1)
    th = _host_threads()
    assert np.array_equal(gpu.ntt(x), cref.ntt(x, threads=th))
    assert np.array_equal(gpu.intt(x), cref.ntt(x, inverse=True, threads=th))


@pytest.mark.skipif(__import__("os").environ.get("MH_NTT") is not None, reason="already inside the re-run")
def test_ntt_large_parity_also_holds_for_the_32_bit_limb_kernel():
    """The same direct comparison at 2^20 with MH_NTT=32, the 32-bit-limb cross-check kernel (the switch is read once per process,
    hence the subprocess)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider", "tests/test_gpu_ntt.py",
                        "-k", "large_matches_the_c_restatement and 20"], cwd=root, env=dict(os.environ, MH_NTT="32"),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-2500:] + r.stderr[-1500:]
