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
