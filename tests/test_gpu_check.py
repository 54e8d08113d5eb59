"""MH_CHECK (marlin_amd/csrc/msm_check.cuh): the invariants of the fixed-base MSM pipeline hold on healthy
batches of every shape -- and each kind of damage a stage can suffer is caught by the check that is there for it, by name.

The function under check replaces VariableBaseMSM::multi_scalar_mul under /root/reference src/lib.rs:172,193,213,292."""
import ctypes as C
import os
import subprocess
import sys
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _report(lib):
    buf = C.create_string_buffer(8192)
    cnt = (C.c_uint64 * 2)()
    assert lib.mh_check_report(buf, 8192, cnt) == 0
    return buf.value.decode(), int(cnt[0]), int(cnt[1])


def _rand_fr(rng, n):
    x = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
    x[:, 3] &= np.uint64((1 << 59) - 1)
    return x


@pytest.fixture
def checked(gpu):
    from marlin_amd import _lib
    lib = _lib.load()
    yield lib
    if hasattr(lib, "mh_debug_corrupt"):
        lib.mh_debug_corrupt(0)
    lib.mh_check_level(int(os.environ.get("MH_CHECK", "0")))


def _hooks_loaded():
    from marlin_amd import _lib
    return os.path.basename(_lib.LIB_PATH).endswith("_testhooks.so")


needs_hooks = pytest.mark.skipif(not _hooks_loaded(), reason="fault injection lives in libmarlin_hip_testhooks.so: run by test_fault_injection_in_the_hooks_library")


@pytest.mark.skipif(_hooks_loaded(), reason="already inside the re-run")
def test_fault_injection_in_the_hooks_library():
    """the product library exports no fault-injection hook: the damage tests below run in a pytest process that loads the hooks
    library (same objects + marlin_amd/csrc/testhooks.hip)"""
    from tests.util import hooks_env
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider", "tests/test_gpu_check.py", "-k", "damage"],
                       cwd=ROOT, env=hooks_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "6 passed" in r.stdout, r.stdout[-3000:] + r.stderr[-1500:]


def test_invariants_hold_on_healthy_batches(gpu, checked):
    """contiguous, aliased (same scalars, shifted bases), strided and bucket-range-sharded batches at both levels"""
    M = gpu
    from marlin_amd import dist as MD, _lib
    rng = np.random.default_rng(5)
    n = 1 << 14
    B = M.Bases.srs_powers(np.array([0x7654321, 0, 0, 0], dtype=np.uint64), n + 40)
    B.precompute(13)
    s1, s2 = _rand_fr(rng, n), _rand_fr(rng, n - 3)
    d1, d2 = M.DeviceBuffer.from_numpy(s1), M.DeviceBuffer.from_numpy(s2)
    ref = M.msm_batch_dev([(B, 0, d1, n), (B, 0, d2, n - 3), (B, 40, d2, n - 3)])
    aff = lambda a: [tuple(M.g1_to_affine(r)[0]) for r in a]
    for level in (1, 2):
        _lib.check(checked.mh_check_level(level), "mh_check_level")
        _, b0, v0 = _report(checked)
        got = M.msm_batch_dev([(B, 0, d1, n), (B, 0, d2, n - 3), (B, 40, d2, n - 3)])      # job 2 shares job 1's lists
        assert aff(got) == aff(ref)
        text, b1, v1 = _report(checked)
        assert b1 == b0 + 1 and v1 == v0, text
        assert "sort: ok" in text and "accumulate: ok" in text and "reduce: ok" in text and "result: ok" in text, text
        if level == 2:
            assert "every entry recodes to its bucket" in text and "recomputed sum of its list" in text and "running-sum reduction" in text, text
        # strided slices (rank 1 of 4)
        l1 = np.ascontiguousarray(s1[1::4])
        e1 = M.DeviceBuffer.from_numpy(l1)
        part = MD.msm_batch_sliced_dev(B, [(1, e1, len(l1))], 4, combine=False)
        Bg = M.Bases(np.ascontiguousarray(B.download()[1:1 + 4 * len(l1):4][:len(l1)]))
        assert aff(part) == aff([M.msm(Bg, l1)])
        text, b2, v2 = _report(checked)
        assert b2 == b1 + 1 and v2 == v0 and "strided 1" in text, text
        e1.free(); Bg.free()
    # bucket-range shards: rank 1 of 2 through the callback transport onto itself (every slot holds this rank's share)
    MD.enable_simulated_shard(1, 2)
    try:
        M.msm_batch_sharded_dev([(B, 0, d1, n)])
        text, _, v3 = _report(checked)
        assert v3 == v0 and "own {1, 2}" in text and "result: ok" in text, text
    finally:
        MD.disable_sharded_prove()


@needs_hooks
@pytest.mark.parametrize("stage,level,names", [(1, 1, "scatter"), (1, 2, "scatter"), (2, 1, "accumulate"), (3, 2, "accumulate"), (4, 2, "reduce + combine")])
def test_each_kind_of_damage_is_caught_by_name(gpu, checked, stage, level, names):
    """mh_debug_corrupt damages the batch's own data after one stage; the call must fail with MH_ECHECK and say where"""
    M = gpu
    from marlin_amd import _lib
    rng = np.random.default_rng(6)
    n = 1 << 13
    B = M.Bases.srs_powers(np.array([0x33221, 0, 0, 0], dtype=np.uint64), n)
    B.precompute(12)
    d = M.DeviceBuffer.from_numpy(_rand_fr(rng, n))
    good = M.msm_batch_dev([(B, 0, d, n)])
    _lib.check(checked.mh_check_level(level), "mh_check_level")
    _, _, v0 = _report(checked)
    _lib.check(checked.mh_debug_corrupt(stage), "mh_debug_corrupt")
    with pytest.raises(_lib.MarlinHipError) as ei:
        M.msm_batch_dev([(B, 0, d, n)])
    assert "code -6" in str(ei.value) and "MH_CHECK" in str(ei.value) and names in str(ei.value), str(ei.value)
    text, _, v1 = _report(checked)
    assert v1 == v0 + 1 and "VIOLATED" in text, text
    # the damage was one-shot: the next batch is checked, passes and gives the right point
    again = M.msm_batch_dev([(B, 0, d, n)])
    assert tuple(M.g1_to_affine(again[0])[0]) == tuple(M.g1_to_affine(good[0])[0])
    assert _report(checked)[2] == v1


@needs_hooks
def test_damage_a_level_cannot_see_is_documented_not_hidden(gpu, checked):
    """a bucket replaced by ANOTHER valid point passes level 1 (every point on the curve, sums consistent) -- level 2 is what
    recomputes buckets; the docstring of msm_check.cuh says so and this pins it"""
    M = gpu
    from marlin_amd import _lib
    rng = np.random.default_rng(7)
    n = 1 << 13
    B = M.Bases.srs_powers(np.array([0x33221, 0, 0, 0], dtype=np.uint64), n)
    B.precompute(12)
    d = M.DeviceBuffer.from_numpy(_rand_fr(rng, n))
    good = M.msm_batch_dev([(B, 0, d, n)])
    _lib.check(checked.mh_check_level(1), "mh_check_level")
    _lib.check(checked.mh_debug_corrupt(3), "mh_debug_corrupt")
    wrong = M.msm_batch_dev([(B, 0, d, n)])
    assert tuple(M.g1_to_affine(wrong[0])[0]) != tuple(M.g1_to_affine(good[0])[0])


def test_a_golden_proof_under_level_2(gpu, checked):
    """a whole proof (DummyCircuit 2^12: 4 fixed-base batches) with every batch checked at level 2, bytes unchanged"""
    from marlin_amd import _lib, marlin as GM
    from oracle import fs as FS
    rng = FS.test_rng()
    a, b = FS.fr_rand(rng), FS.fr_rand(rng)
    n = 1 << 12
    srs = GM.universal_setup(n, n, 3 * n, 0x123456789abcdef, 0xfedcba987654321)
    ncp, ni, mats, inst, wit = GM.dummy_circuit(a, b, 10, n)
    pk = GM.index(srs, ncp, ni, mats)
    want = GM.prove(pk, inst, wit, bytes(range(32)))
    _lib.check(checked.mh_check_level(2), "mh_check_level")
    _, b0, v0 = _report(checked)
    got = GM.prove(pk, inst, wit, bytes(range(32)))
    text, b1, v1 = _report(checked)
    assert got == want and b1 >= b0 + 4 and v1 == v0, text


def test_soak_tool_short_run(gpu, tmp_path):
    """tools/soak_sliced.py -- the 8-process scenario of the one unexplained failure, and the same MSMs sharded by bucket range (the
    accumulate kernel's cut buckets) -- for a few iterations at level 2 (<= 60 s)"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak_sliced.py"), "--world", "8", "--iters", "12", "--check", "2",
                        "--sharded-c", "16", "--ntt-logs", "6", "13", "--out", str(tmp_path), "--port", "29877", "--timeout", "300"],
                       capture_output=True, text=True, timeout=400)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("SOAK ")][-1]
    import json
    tot = json.loads(line[5:])
    assert tot["mismatches"] == 0 and tot["errors"] == 0 and tot["violations"] == 0 and tot["sliced_msms"] == 8 * 12 * 6 and tot["bucket_range_sharded_msms"] == 8 * 12 * 3 and tot["batches_checked"] >= 8 * 12 * 4, line
