"""Pins the C restatement (oracle/c/ref_hotpath.c) against the pure-Python oracle:
naive DFT / radix-2 NTT, naive MSM, known-dlog identities.  CPU only."""
import numpy as np
from oracle import fields as F, curve as EC, poly as OP, cref
from tests.util import fr_to_np, np_to_fr, points_to_np, jac_np_to_affine, arith_bases, rand_fr, limbs_to_fq


def test_c_ntt_matches_python():
    for log_n in range(0, 11):
        v = rand_fr(1 << log_n, log_n)
        assert np_to_fr(cref.ntt(fr_to_np(v))) == OP.ntt(v, log_n)
        assert np_to_fr(cref.ntt(fr_to_np(v), inverse=True)) == OP.ntt(v, log_n, inverse=True)
    v = rand_fr(32, 99)
    assert np_to_fr(cref.ntt(fr_to_np(v))) == OP.dft_naive(v, 5)
    # the OpenMP-threaded stages (cpu_baseline's all-core figure) give the same transform, chunked twiddle table included
    rng = np.random.default_rng(3)
    x = rng.integers(0, 1 << 63, size=(1 << 14, 4), dtype=np.uint64)
    x[:, 3] &= np.uint64((1 << 61) - 1)
    for inv in (False, True):
        assert np.array_equal(cref.ntt(x, inverse=inv, threads=4), cref.ntt(x, inverse=inv, threads=1))


def test_c_bases_and_mul_gen():
    pts, dl = arith_bases(40)
    got, dl2 = cref.bases_arith(40)
    assert dl == dl2
    assert np.array_equal(got, points_to_np(pts))
    for k in [0, 1, 2, F.R_MOD - 1, 0xdeadbeef12345]:
        assert jac_np_to_affine(cref.g1_mul_gen(k)) == EC.scalar_mul(EC.G1_GEN, k)


def test_c_msm_matches_python():
    pts, dl = arith_bases(64)
    b = points_to_np(pts)
    for n in [1, 3, 31, 32, 64]:
        sc = rand_fr(n, n)
        want = EC.msm_naive(pts[:n], sc)
        assert jac_np_to_affine(cref.msm(b[:n], fr_to_np(sc))) == want
        assert jac_np_to_affine(cref.msm(b[:n], fr_to_np(sc, montgomery=False), montgomery=False, threads=4)) == want
    # edge scalars incl. the scalar==1 shortcut and zeros
    sc = [0, 1, 1, F.R_MOD - 1, 2, 0, 1 << 254, 5] * 8
    assert jac_np_to_affine(cref.msm(b, fr_to_np(sc))) == EC.msm_naive(pts, sc)
    xy, inf = cref.g1_to_affine(cref.msm(b, fr_to_np([0] * 64)))
    assert inf


def test_c_msm_known_dlog_large():
    n = 20000
    b, dl = cref.bases_arith(n)
    sc = rand_fr(n, 5)
    k = sum(s * a for s, a in zip(sc, dl)) % F.R_MOD
    out = cref.msm(b, fr_to_np(sc), threads=8)
    assert jac_np_to_affine(out) == EC.scalar_mul(EC.G1_GEN, k)


def test_c_poly_helpers_match_python():
    """the dense-polynomial helpers behind the large-size opening pins (tests/test_gpu_parity_pins.py): linear
    combination, synthetic division by (X - z), Horner evaluation, shifted accumulation"""
    a, b = rand_fr(50, 1), rand_fr(30, 2)
    c1, c2 = 5, F.R_MOD - 3
    got = np_to_fr(cref.lincomb([(c1, fr_to_np(a)), (c2, fr_to_np(b))], threads=2))
    assert got == [(c1 * a[i] + c2 * (b[i] if i < 30 else 0)) % F.R_MOD for i in range(50)]
    assert np_to_fr(cref.lincomb([(c1, fr_to_np(a))], n=20)) == [c1 * x % F.R_MOD for x in a[:20]]
    for z in (0, 1, 123456789, F.R_MOD - 1):
        q, rem = cref.div_linear(fr_to_np(a), z)
        assert np_to_fr(q) == OP.divide_by_linear(a, z)
        assert rem == OP.poly_eval(a, z) == cref.poly_eval(fr_to_np(a), z)
    q, rem = cref.div_linear(fr_to_np(a[:1]), 7)
    assert len(q) == 0 and rem == a[0]
    dst = fr_to_np(a)
    cref.add_at(dst, 20, fr_to_np(b))
    assert np_to_fr(dst) == [(a[i] + (b[i - 20] if i >= 20 else 0)) % F.R_MOD for i in range(50)]


def test_c_restatement_on_bn254():
    """the same pins for the BN254 build (libref_hotpath_bn254.so; BASELINE.json configs[4]); the oracle picks its curve
    at import, so the checks run in a subprocess"""
    import os, subprocess, sys
    if F.CURVE == "bn254":
        return
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", "tests/test_oracle_c.py"], cwd=root,
                       env=dict(os.environ, ORACLE_CURVE="bn254"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
