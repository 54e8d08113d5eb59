"""Parity of the HIP Pippenger MSM (through the C ABI) against the oracle.
Bit-exact after affine normalisation (the form the reference serialises/hashes)."""
import os
import numpy as np
import os
import pytest
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
from oracle import fields as F
from oracle import curve as EC
from tests.util import fr_to_np, points_to_np, jac_np_to_affine, arith_bases, rand_fr, limbs_to_fq

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def bases4k():
    pts, dl = arith_bases(4096)
    return pts, dl


def _check_dlog(gpu, B, pts, dl, scalars, offset=0, montgomery=True):
    n = len(scalars)
    out = gpu.msm(B, fr_to_np(scalars, montgomery=montgomery), base_offset=offset, montgomery=montgomery)
    got = jac_np_to_affine(out)
    k = sum(s * a for s, a in zip(scalars, dl[offset:offset + n])) % F.R_MOD
    want = EC.scalar_mul(EC.G1_GEN, k)
    assert got == want
    assert EC.is_on_curve(got)
    # host normalisation helper agrees with the oracle's
    xy, inf = gpu.g1_to_affine(out)
    if want is None:
        assert inf
    else:
        assert not inf and (limbs_to_fq(xy[:F.FQ_LIMBS64]), limbs_to_fq(xy[F.FQ_LIMBS64:])) == want


@pytest.mark.parametrize("n", [1, 2, 3, 5, 31, 32, 33, 100, 257])
def test_msm_small_vs_naive(gpu, bases4k, n):
    pts, dl = bases4k
    B = gpu.Bases(points_to_np(pts[:n]))
    sc = rand_fr(n, n)
    out = gpu.msm(B, fr_to_np(sc))
    assert jac_np_to_affine(out) == EC.msm_naive(pts[:n], sc)
    if n <= 100:
        assert jac_np_to_affine(out) == EC.msm_pippenger(pts[:n], sc)


@pytest.mark.parametrize("n", [1000, 4096])
def test_msm_known_dlog(gpu, bases4k, n):
    pts, dl = bases4k
    B = gpu.Bases(points_to_np(pts[:n]))
    _check_dlog(gpu, B, pts, dl, rand_fr(n, 77 + n))


@pytest.mark.parametrize("n", [2 ** 13 + 5, 2 ** 14, 2 ** 15 - 1, 2 ** 16, 2 ** 17 + 3, 2 ** 18])
def test_msm_every_window_width(gpu, bases4k, n):
    """n chosen so the plan uses c = 9 ... 13 (mixed c / c-1 wide windows); bases are the
    4096 known-dlog points tiled, so the answer is one scalar multiplication."""
    pts, dl = bases4k
    reps = (n + 4095) // 4096
    big = np.tile(points_to_np(pts), (reps, 1))[:n]
    B = gpu.Bases(big)
    sc = rand_fr(n, n)
    out = gpu.msm(B, fr_to_np(sc))
    k = sum(s * dl[i % 4096] for i, s in enumerate(sc)) % F.R_MOD
    assert jac_np_to_affine(out) == EC.scalar_mul(EC.G1_GEN, k)


def test_msm_offsets_and_canonical_scalars(gpu, bases4k):
    pts, dl = bases4k
    B = gpu.Bases(points_to_np(pts))
    _check_dlog(gpu, B, pts, dl, rand_fr(700, 5), offset=1234)
    _check_dlog(gpu, B, pts, dl, rand_fr(300, 6), offset=3796)          # ends exactly at the last base
    _check_dlog(gpu, B, pts, dl, rand_fr(513, 8), offset=11, montgomery=False)


def test_msm_edge_scalars(gpu, bases4k):
    pts, dl = bases4k
    n = 600
    B = gpu.Bases(points_to_np(pts[:n]))
    r = F.R_MOD
    _check_dlog(gpu, B, pts, dl, [0] * n)                      # identity result
    _check_dlog(gpu, B, pts, dl, [1] * n)
    _check_dlog(gpu, B, pts, dl, [r - 1] * n)
    _check_dlog(gpu, B, pts, dl, [0] * (n - 1) + [5])
    _check_dlog(gpu, B, pts, dl, [(1 << 254) + 12345] * n)     # high bit set: top-window carry
    _check_dlog(gpu, B, pts, dl, [2 ** 15] * n)                # digit exactly half a window
    _check_dlog(gpu, B, pts, dl, [2 ** 16 - 1] * n)
    # scalars that cancel: s*P + (r-s)*P = O
    sc = rand_fr(n // 2, 3)
    B2 = gpu.Bases(points_to_np([p for p in pts[:n // 2] for _ in (0, 1)]))
    out = gpu.msm(B2, fr_to_np([v for s in sc for v in (s, r - s)]))
    assert jac_np_to_affine(out) is None


def test_msm_repeated_base(gpu, bases4k):
    """every base identical: every bucket addition after the first hits the doubling branch."""
    pts, dl = bases4k
    n = 500
    B = gpu.Bases(points_to_np([pts[7]] * n))
    sc = rand_fr(n, 21)
    out = gpu.msm(B, fr_to_np(sc))
    assert jac_np_to_affine(out) == EC.scalar_mul(pts[7], sum(sc) % F.R_MOD)
    # equal scalars too: one bucket per window holds all n points
    out = gpu.msm(B, fr_to_np([sc[0]] * n))
    assert jac_np_to_affine(out) == EC.scalar_mul(pts[7], sc[0] * n % F.R_MOD)


def test_msm_empty_and_errors(gpu, bases4k):
    pts, dl = bases4k
    B = gpu.Bases(points_to_np(pts[:8]))
    out = gpu.msm(B, np.zeros((0, 4), dtype=np.uint64))
    assert jac_np_to_affine(out) is None
    with pytest.raises(gpu.MarlinHipError):
        gpu.msm(B, fr_to_np([1] * 9))                      # more scalars than bases
    with pytest.raises(gpu.MarlinHipError):
        gpu.msm(B, fr_to_np([1] * 4), base_offset=6)


def test_bases_upload_rejects_points_off_the_curve(gpu, bases4k):
    """An affine point has no infinity flag on this ABI (ark's VariableBaseMSM accepts the identity as a base): a base set
    containing an identity encoded as (0, 0) or (0, 1), or any other off-curve point, is refused at upload instead of
    silently yielding a wrong MSM."""
    pts, dl = bases4k
    arr = points_to_np(pts[:16])
    for bad_xy in ((0, 0), (0, 1), (pts[3][0], pts[4][1])):
        a = arr.copy()
        a[5] = points_to_np([bad_xy])[0]
        with pytest.raises(gpu.MarlinHipError, match="not on the curve"):
            gpu.Bases(a)
    gpu.Bases(arr)                                         # the clean set uploads


def test_msm_skewed_buckets_use_pair_tree(gpu, bases4k):
    """2^16 points, only 3 distinct scalars: a handful of buckets hold ~all points (the thread-per-bucket loop would
    serialise tens of thousands of additions; the driver switches to the pair-tree accumulation)."""
    import time
    pts, dl = bases4k
    n = 1 << 16
    big = np.tile(points_to_np(pts), (n // 4096, 1))
    B = gpu.Bases(big)
    vals = rand_fr(3, 31)
    sc = [vals[i % 3] for i in range(n)]
    t0 = time.time()
    out = gpu.msm(B, fr_to_np(sc))
    dt = time.time() - t0
    k = sum(s * dl[i % 4096] for i, s in enumerate(sc)) % F.R_MOD
    assert jac_np_to_affine(out) == EC.scalar_mul(EC.G1_GEN, k)
    assert dt < 5.0


def test_msm_batch_equals_individual(gpu, bases4k):
    """mh_msm_batch_dev: jobs of very different sizes, offsets and an empty job give the same group elements as
    one-at-a-time calls (and the known-dlog answer)."""
    pts, dl = bases4k
    B = gpu.Bases(points_to_np(pts))
    sizes = [4096, 3, 0, 1000, 2500, 17, 4000, 1, 333, 2048]          # 10 jobs > MAX_JOBS: exercises the grouping
    offs = [0, 5, 0, 3096, 100, 4079, 96, 4095, 1, 2048]
    bufs, jobs, want = [], [], []
    for k, (n, off) in enumerate(zip(sizes, offs)):
        sc = rand_fr(n, 1000 + k)
        buf = gpu.DeviceBuffer.from_numpy(fr_to_np(sc)) if n else gpu.DeviceBuffer(32)
        bufs.append(buf)
        jobs.append((B, off, buf, n))
        want.append(EC.scalar_mul(EC.G1_GEN, sum(s * a for s, a in zip(sc, dl[off:off + n])) % F.R_MOD) if n else None)
    out = gpu.msm_batch_dev(jobs)
    for k in range(len(jobs)):
        assert jac_np_to_affine(out[k]) == want[k], k
        if sizes[k]:
            assert jac_np_to_affine(gpu.msm_dev(B, bufs[k], sizes[k], base_offset=offs[k])) == want[k]


# ---- fixed-base path: mh_bases_precompute + Pippenger over the window table (msm_fb.cuh) -----------------------

@pytest.mark.parametrize("cbits,n", [(6, 4096), (9, 2 ** 14), (13, 2 ** 16 + 7), (16, 2 ** 17), (17, 2 ** 18), (18, 2 ** 19 - 3), (20, 2 ** 19 + 11)])
def test_msm_fixed_base_every_layout(gpu, bases4k, cbits, n):
    """window widths with one virtual window (c <= 13) and with 8 ... 128 of them (c = 16 ... 20); known-dlog answer, and the
    same element as the variable-base path on a set without a table."""
    pts, dl = bases4k
    reps = (n + 4095) // 4096
    big = np.tile(points_to_np(pts), (reps, 1))[:n]
    B = gpu.Bases(big).precompute(cbits)
    sc = rand_fr(n, 900 + n)
    fb0, vb0 = gpu.msm_path_counts()
    out = gpu.msm(B, fr_to_np(sc))
    assert gpu.msm_path_counts() == (fb0 + 1, vb0)            # served by the fixed-base path
    k = sum(s * dl[i % 4096] for i, s in enumerate(sc)) % F.R_MOD
    assert jac_np_to_affine(out) == EC.scalar_mul(EC.G1_GEN, k)
    B0 = gpu.Bases(big)
    assert jac_np_to_affine(gpu.msm(B0, fr_to_np(sc))) == jac_np_to_affine(out)
    assert gpu.msm_path_counts() == (fb0 + 1, vb0 + 1)


def test_msm_fixed_base_offsets_batch_and_edges(gpu, bases4k):
    """offsets into the tabled set, canonical scalars, a batch of jobs (one empty, one tiny; 9 live jobs = two groups),
    edge scalars 0 / 1 / r-1 / 2^k with heavily repeated (base, scalar) pairs (the deferred-collision fix-up), and a
    skewed input that makes the driver fall back to the variable-base path and its pair tree."""
    pts, dl = bases4k
    n = 1 << 15
    big = np.tile(points_to_np(pts), (n // 4096, 1))
    dlb = [dl[i % 4096] for i in range(n)]
    B = gpu.Bases(big).precompute(10)
    r = F.R_MOD

    def want(sc, off):
        return EC.scalar_mul(EC.G1_GEN, sum(s * a for s, a in zip(sc, dlb[off:off + len(sc)])) % r)

    sc = rand_fr(20000, 3)
    assert jac_np_to_affine(gpu.msm(B, fr_to_np(sc), base_offset=12345)) == want(sc, 12345)
    assert jac_np_to_affine(gpu.msm(B, fr_to_np(sc[:12768]), base_offset=20000)) == want(sc[:12768], 20000)   # ends at the last base
    assert jac_np_to_affine(gpu.msm(B, fr_to_np(sc, montgomery=False), base_offset=1, montgomery=False)) == want(sc, 1)
    edge = ([0, 1, r - 1, 2, r - 2, 1 << 254, (1 << 255) % r, 1 << 19, (1 << 20) - 1, 1 << 9, 513] * 2000)[:n - 5]
    assert jac_np_to_affine(gpu.msm(B, fr_to_np(edge), base_offset=5)) == want(edge, 5)
    assert jac_np_to_affine(gpu.msm(B, fr_to_np([0] * 9000))) is None
    sizes = [30000, 0, 12000, 40, 32768, 9000, 25000, 8191, 16384, 31000]
    offs = [100, 0, 20000, 7, 0, 1, 7000, 3, 16384, 1768]
    bufs, jobs, exp = [], [], []
    for k, (m, off) in enumerate(zip(sizes, offs)):
        s = rand_fr(m, 2000 + k)
        bufs.append(gpu.DeviceBuffer.from_numpy(fr_to_np(s)) if m else gpu.DeviceBuffer(32))
        jobs.append((B, off, bufs[-1], m))
        exp.append(want(s, off) if m else None)
    out = gpu.msm_batch_dev(jobs)
    for k in range(len(jobs)):
        assert jac_np_to_affine(out[k]) == exp[k], k
    vals = rand_fr(3, 41)
    skew = [vals[i % 3] for i in range(n)]
    assert jac_np_to_affine(gpu.msm(B, fr_to_np(skew))) == want(skew, 0)
    B16 = gpu.Bases(big).precompute(16)          # 2^15 buckets, average load 16, largest ~10900: falls back
    fb0, vb0 = gpu.msm_path_counts()
    assert jac_np_to_affine(gpu.msm(B16, fr_to_np(skew))) == want(skew, 0)
    assert gpu.msm_path_counts() == (fb0, vb0 + 1)
    with pytest.raises(gpu.MarlinHipError):
        gpu.Bases(big[:64]).precompute(23)


def test_fixed_base_table_info_and_shard_resize(gpu, bases4k):
    """automatic window width (lg(n) below 2^19 points, lg(n) - 1 above, <= 20), 128-byte table points; mh_marlin_set_shard leaves the tables alone
    (the prover shards by bucket range, the window width does not depend on the number of ranks) and a plain mh_msm
    is never sharded: results do not change."""
    import ctypes as C
    from marlin_amd import _lib
    pts, dl = bases4k
    n = 1 << 15
    big = np.tile(points_to_np(pts), (n // 4096, 1))
    dlb = [dl[i % 4096] for i in range(n)]
    B = gpu.Bases(big)
    assert B.table_info() == (0, 0, 0)
    B.precompute()
    c, w, nbytes = B.table_info()
    pt = 128 if F.FQ_LIMBS64 == 6 else 96
    from tests.util import auto_window_bits
    c0 = auto_window_bits(n)
    assert c0 == 15 and (c, w, nbytes) == (c0, (256 + c0 - 1) // c0, ((256 + c0 - 1) // c0) * n * pt)
    Bx = gpu.Bases(big).precompute(9)
    sc = rand_fr(n, 4242)
    want = EC.scalar_mul(EC.G1_GEN, sum(s * a for s, a in zip(sc, dlb)) % F.R_MOD)
    assert jac_np_to_affine(gpu.msm(B, fr_to_np(sc))) == want
    L = _lib.load()
    cb_t = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p)
    cb = cb_t(lambda *a: -1)                       # never called: plain mh_msm does not shard
    try:
        _lib.check(L.mh_marlin_set_shard(0, 4, C.cast(cb, C.c_void_p), None), "mh_marlin_set_shard")
        assert B.table_info()[0] == c0 and Bx.table_info()[0] == 9
        assert jac_np_to_affine(gpu.msm(B, fr_to_np(sc))) == want
    finally:
        _lib.check(L.mh_marlin_set_shard(0, 1, None, None), "mh_marlin_set_shard")
    assert B.table_info()[0] == c0


def test_msm_fixed_base_jobs_sharing_scalars(gpu, bases4k):
    """Jobs of one batch that multiply the SAME scalar vector against ranges of the same tabled base set (what
    MarlinKZG10::commit does for a degree-bounded polynomial: powers and shifted_powers(d)) share one sort: results equal
    the known-dlog answers, including with repeated (base, scalar) pairs (the equal-x buckets are recomputed by the
    fix-up pass, whose lists are shared) and with the offsets in either order."""
    pts, dl = bases4k
    n = 1 << 15
    big = np.tile(points_to_np(pts), (n // 4096, 1))
    dlb = [dl[i % 4096] for i in range(n)]
    B = gpu.Bases(big).precompute(12)
    r = F.R_MOD

    def want(sc, off):
        return EC.scalar_mul(EC.G1_GEN, sum(s * a for s, a in zip(sc, dlb[off:off + len(sc)])) % r)

    m = 20000
    sc = rand_fr(m, 99)
    buf = gpu.DeviceBuffer.from_numpy(fr_to_np(sc))
    other = rand_fr(9000, 100)
    obuf = gpu.DeviceBuffer.from_numpy(fr_to_np(other))
    fb0, _ = gpu.msm_path_counts()
    out = gpu.msm_batch_dev([(B, 0, buf, m), (B, 4096 + 7, buf, m), (B, 11, obuf, 9000), (B, 3, buf, m)])
    assert gpu.msm_path_counts()[0] > fb0
    assert [jac_np_to_affine(o) for o in out] == [want(sc, 0), want(sc, 4103), want(other, 11), want(sc, 3)]
    out = gpu.msm_batch_dev([(B, 500, buf, m), (B, 2, buf, m)])                      # the later job starts EARLIER: no sharing
    assert [jac_np_to_affine(o) for o in out] == [want(sc, 500), want(sc, 2)]
    # repeated pairs: base i and base i + 4096 are the same point, equal scalars put them into the same buckets
    rep = ([5, 5, r - 5, 77] * (m // 4))[:m]
    rbuf = gpu.DeviceBuffer.from_numpy(fr_to_np(rep))
    out = gpu.msm_batch_dev([(B, 0, rbuf, m), (B, 4096, rbuf, m), (B, 1, rbuf, m)])
    assert [jac_np_to_affine(o) for o in out] == [want(rep, 0), want(rep, 4096), want(rep, 1)]


def test_fq30_device_selftest(gpu):
    """the 30-bit-limb field / group arithmetic of the fixed-base path agrees with the 32-bit Montgomery arithmetic on
    2^18 pseudo-random operand pairs plus structured ones (mh_selftest_fq30: a hook of libmarlin_hip_testhooks.so -- the product's
    objects + testhooks.hip --, so it runs in a process that loads that library)."""
    import subprocess, sys
    from tests.util import hooks_env
    code = ("import ctypes as C, marlin_amd as M\nfrom marlin_amd import _lib\nM.init(0)\n"
            "for seed in (1, 2):\n    bad = C.c_uint64(123)\n    _lib.check(_lib.load().mh_selftest_fq30(1 << 18, seed, C.byref(bad)), 'selftest')\n"
            "    assert bad.value == 0, bad.value\nprint('selftest ok')\n")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=hooks_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "selftest ok" in r.stdout, r.stdout[-1000:] + r.stderr[-2000:]


def test_fixed_base_msm_at_2p22_matches_the_c_restatement(gpu):
    """the fixed-base MSM kernels at the size of the largest commitment of a 2^20-constraint proof -- 2^22
    pseudo-random points ([tau^i]G, generated on the device and downloaded), window table at c = 20, one bucket set of 2^19, the
    row / column + bit-plane reduction at full size -- against the C restatement's Pippenger (oracle/c/ref_hotpath.c: unsigned
    windows, one bucket set per window: another algorithm, the same group element) on the same points and scalars, and against the
    closed form [p(tau)]G.  `VariableBaseMSM::multi_scalar_mul` as reached from /root/reference src/lib.rs:172,193,213,292."""
    import os
    from oracle import cref, poly as OP, curve as EC
    n = 1 << 22
    tau = 0x1f3a9c5d7e2b4a6f8091a2b3c4d5e6f7
    B = gpu.Bases.srs_powers(fr_to_np([tau])[0], n)
    B.precompute()
    c, W, _ = B.table_info()
    assert (c, W) == (20, 13)
    rng = np.random.default_rng(22)
    sc = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)        # Montgomery words of pseudo-random field elements
    sc[:, 3] &= np.uint64((1 << 61) - 1)
    sc[:5] = 0                                                          # a few zeros and a repeated scalar among them
    sc[5:9] = sc[9]
    fb0, _ = gpu.msm_path_counts()
    got = jac_np_to_affine(gpu.msm(B, sc))
    assert gpu.msm_path_counts()[0] > fb0                               # the fixed-base path answered
    try:
        th = max(1, min(16, len(os.sched_getaffinity(0))))
    except AttributeError:
        th = 4
    assert got == jac_np_to_affine(cref.msm(B.download(), sc, threads=th))
    # and the closed form: sum_i s_i tau^i by Horner over the canonical scalars (numpy object arithmetic would take minutes:
    # the C restatement's poly_eval on the same Montgomery words)
    assert got == EC.scalar_mul(EC.G1_GEN, cref.poly_eval(sc, tau))
