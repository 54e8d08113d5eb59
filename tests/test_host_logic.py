"""CPU tests of the product's host-side logic (marlin_amd/csrc/host_ff.h, fs_host.h) through the
host-only hook library libmarlin_hosttest.so, against hashlib and the oracle."""
import ctypes as C
import hashlib
import os
import numpy as np
from oracle import fields as F, curve as EC, fs as FS, marlin as MR
from tests.util import fr_to_np, np_to_fr, fq_to_limbs, limbs_to_fq

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = C.CDLL(os.path.join(ROOT, "marlin_amd", "libmarlin_hosttest.so"))


def test_blake2s_matches_hashlib():
    for n in [0, 1, 3, 63, 64, 65, 127, 128, 129, 1000]:
        data = bytes((i * 7 + n) & 0xFF for i in range(n))
        out = (C.c_uint8 * 32)()
        L.ht_blake2s(data, C.c_size_t(n), out)
        assert bytes(out) == hashlib.blake2s(data).digest(), n


def test_chacha_and_fr_rand_match_oracle():
    seed = bytes(range(32))
    for rounds in (20, 12):
        out = np.zeros(200, dtype=np.uint64)
        L.ht_chacha_u64(seed, rounds, C.c_size_t(200), C.c_void_p(out.ctypes.data))
        r = FS.ChaChaRng(seed, rounds)
        assert [int(x) for x in out] == [r.next_u64() for _ in range(200)]
    out = np.zeros((50, 4), dtype=np.uint64)
    L.ht_fr_rand(seed, 20, C.c_size_t(50), C.c_void_p(out.ctypes.data))
    r = FS.ChaChaRng(seed, 20)
    assert np_to_fr(out) == [FS.fr_rand(r) for _ in range(50)]


def test_fiat_shamir_matches_oracle():
    init, a1 = b"MARLIN-2019" + bytes(range(100)), bytes(range(200, 256)) * 7
    out = np.zeros((5, 4), dtype=np.uint64)
    L.ht_fs(init, C.c_size_t(len(init)), a1, C.c_size_t(len(a1)), C.c_void_p(out.ctypes.data))
    fs = FS.SimpleHashFiatShamirRng(init)
    fs.absorb(a1)
    want = [fs.rand_fr() for _ in range(4)] + [fs.rand_u128_as_fr()]
    assert np_to_fr(out) == want


def test_host_field_and_group_ops():
    a, b = 0x1234567890abcdef1234567890abcdef1234567890abcdef, F.R_MOD - 5
    x, y = fr_to_np([a])[0], fr_to_np([b])[0]
    r = np.zeros(4, dtype=np.uint64)
    L.ht_fr_mul(C.c_void_p(x.ctypes.data), C.c_void_p(y.ctypes.data), C.c_void_p(r.ctypes.data))
    assert np_to_fr(r)[0] == a * b % F.R_MOD
    L.ht_fr_inv(C.c_void_p(x.ctypes.data), C.c_void_p(r.ctypes.data))
    assert np_to_fr(r)[0] == pow(a, -1, F.R_MOD)
    qa, qb = F.Q_MOD - 12345, 0xdeadbeef << 300
    xa = np.array(fq_to_limbs(qa), dtype=np.uint64); xb = np.array(fq_to_limbs(qb), dtype=np.uint64)
    rq = np.zeros(6, dtype=np.uint64)
    L.ht_fq_mul(C.c_void_p(xa.ctypes.data), C.c_void_p(xb.ctypes.data), C.c_void_p(rq.ctypes.data))
    assert limbs_to_fq(rq) == qa * qb % F.Q_MOD
    g = np.array(fq_to_limbs(EC.G1_GEN[0]) + fq_to_limbs(EC.G1_GEN[1]), dtype=np.uint64)
    for k in [1, 2, 0xabcdef123456789, F.R_MOD - 1]:
        kk = np.array(F.to_limbs64(k, 4), dtype=np.uint64)
        out = np.zeros(12, dtype=np.uint64); inf = C.c_int()
        L.ht_g1_mul(C.c_void_p(g.ctypes.data), C.c_void_p(kk.ctypes.data), C.c_void_p(out.ctypes.data), C.byref(inf))
        assert (limbs_to_fq(out[:6]), limbs_to_fq(out[6:])) == EC.scalar_mul(EC.G1_GEN, k)
    # commitment ToBytes layout == oracle's
    p2 = EC.scalar_mul(EC.G1_GEN, 77)
    xy2 = np.array(fq_to_limbs(p2[0]) + fq_to_limbs(p2[1]), dtype=np.uint64)
    out = (C.c_uint8 * 195)()
    L.ht_put_commitment(C.c_void_p(g.ctypes.data), 0, None, out)
    assert bytes(out) == MR.commitment_bytes((EC.G1_GEN, None))
    L.ht_put_commitment(C.c_void_p(g.ctypes.data), 1, C.c_void_p(xy2.ctypes.data), out)
    assert bytes(out) == MR.commitment_bytes((EC.G1_GEN, (p2,)))
