"""Contexts and groups: the library's state lives in context handles instead of process globals, so

* two contexts on ONE GPU prove at the same time from two threads and both give the serial bytes;
* one process drives N ranks -- a group of contexts joined by the in-process transport -- and every rank's proof is the one-GPU
  proof, at 2 / 3 / 4 / 8 ranks (sliced rounds from 4 on), from Python threads and from the C example `examples/prove_multi_rank.c`,
  which also forks N processes over the native RCCL transport (the stand-in for librccl, synchronous and enqueued).

The reference shape is one caller and rayon threads inside one Marlin::prove (/root/reference src/lib.rs:151-155, src/ahp/mod.rs:9-10)."""
import ctypes as C
import os
import subprocess
import sys
import threading
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MOCK = os.path.join(ROOT, "tests", "mock_rccl", "libmock_rccl.so")
TAU, GAMMA = 0x123456789abcdef123, 0xfedcba987654321f
SEED = bytes(range(32))


def _setup_index_prove(log_n, proofs=1):
    """universal_setup -> index -> prove on the calling thread's CURRENT context"""
    from marlin_amd import marlin as GM
    from oracle import fs as FS
    rng = FS.test_rng()
    a, b = FS.fr_rand(rng), FS.fr_rand(rng)
    n = 1 << log_n
    srs = GM.universal_setup(n, n, 3 * n, TAU, GAMMA)
    ncp, ni, mats, inst, wit = GM.dummy_circuit(a, b, 10, n)
    pk = GM.index(srs, ncp, ni, mats)
    out = [GM.prove(pk, inst, wit, SEED) for _ in range(proofs)]
    pk.free()
    return out


def test_two_contexts_on_one_gpu_prove_concurrently(gpu):
    from marlin_amd import _lib
    lib = _lib.load()
    want = _setup_index_prove(13)[0]                                  # the default context, alone
    ctxs = []
    for _ in range(2):
        h = C.c_void_p()
        _lib.check(lib.mh_ctx_create(0, C.byref(h)), "mh_ctx_create")
        ctxs.append(h)
    got, errs = [None, None], []
    gate = threading.Barrier(2)

    def work(i):
        try:
            _lib.check(lib.mh_ctx_set_current(ctxs[i]), "mh_ctx_set_current")
            assert lib.mh_ctx_get_current() == ctxs[i].value
            gate.wait(timeout=120)
            got[i] = _setup_index_prove(13, proofs=6)                 # keys, tables and workspaces of its own; six proofs side by side
        except Exception as e:                                         # noqa: BLE001
            errs.append("context %d: %r" % (i, e))
        finally:
            lib.mh_ctx_set_current(None)
    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    [t.start() for t in th]
    [t.join(600) for t in th]
    assert not errs, errs
    assert all(p == want for g in got for p in g)
    # handles belong to their context: the default context does not know a key or a base set of another one, and still works
    assert _setup_index_prove(13)[0] == want
    for h in ctxs:
        _lib.check(lib.mh_ctx_destroy(h), "mh_ctx_destroy")
    assert lib.mh_ctx_destroy(ctxs[0]) != 0                           # unknown by now


@pytest.mark.parametrize("world,log_n", [(2, 12), (3, 12), (4, 12), (8, 13)])
def test_group_of_contexts_gives_the_single_gpu_proof(gpu, world, log_n):
    """mh_group_create + mh_group_run from Python: rank r's work runs on a library thread bound to context r"""
    from marlin_amd import _lib, dist as MD
    lib = _lib.load()
    want = _setup_index_prove(log_n)[0]
    grp = C.c_void_p()
    devs = (C.c_int * world)(*([0] * world))
    _lib.check(lib.mh_group_create(devs, world, C.byref(grp)), "mh_group_create")
    assert lib.mh_group_size(grp) == world and lib.mh_group_ctx(grp, world - 1) and not lib.mh_group_ctx(grp, world)
    proofs, stats = {}, {}

    @C.CFUNCTYPE(C.c_int, C.c_int, C.c_void_p)
    def rank_main(rank, _user):
        try:
            proofs[rank] = _setup_index_prove(log_n, proofs=2)
            stats[rank] = MD.exchange_stats()
            return 0
        except Exception as e:                                         # noqa: BLE001
            proofs[rank] = repr(e)
            return -3
    rc = lib.mh_group_run(grp, rank_main, None)
    assert rc == 0, (rc, lib.mh_last_error(), proofs)
    assert all(proofs[r] == [want, want] for r in range(world)), {r: (p if isinstance(p, str) else [x == want for x in p]) for r, p in proofs.items()}
    # the rounds really were sharded: 4 all-gathers of partial points per proof, and from 4 ranks on the sliced sections' exchanges
    per_proof = stats[0][0] / 2
    assert per_proof >= (12 if world in (4, 8) else 4), stats
    _lib.check(lib.mh_group_destroy(grp), "mh_group_destroy")


def _build(tmp_path):
    exe = str(tmp_path / "prove_multi_rank")
    lib_dir = os.path.join(ROOT, "marlin_amd")
    r = subprocess.run(["gcc", "-std=c99", "-O1", "-Wall", "-Wextra", "-pedantic", "-Werror", "-pthread", "-I" + os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "examples", "prove_multi_rank.c"), "-L" + lib_dir, "-lmarlin_hip", "-Wl,-rpath," + lib_dir, "-o", exe],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


@pytest.mark.parametrize("mode,ranks,enqueued", [("threads", 2, False), ("threads", 4, False), ("fork", 2, False), ("fork", 4, False), ("fork", 4, True)])
def test_c_example_multi_rank_without_python(gpu, tmp_path, mode, ranks, enqueued):
    """examples/prove_multi_rank.c, built -Werror: N ranks through the C ABI alone -- threads over a group of contexts, or forked
    processes over the library's own RCCL communicator (ncclUniqueId through pipes) -- and the parent compares every rank's proof
    with the one-rank proof; exit status 0 = identical"""
    exe = _build(tmp_path)
    env = dict(os.environ)
    if mode == "fork":
        env["MH_RCCL_LIB"] = MOCK
        if enqueued:
            env["MH_MOCK_RCCL_ASYNC"] = "1"
    r = subprocess.run([exe, "--mode", mode, "--ranks", str(ranks), "--devices", ",".join(["0"] * ranks), "--log", "12"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "equals the one-rank proof" in r.stdout, r.stdout + r.stderr


def test_c_example_reports_a_failing_rank(gpu, tmp_path):
    """a rank that cannot even initialise (device 99) makes the program exit non-zero instead of hanging"""
    exe = _build(tmp_path)
    r = subprocess.run([exe, "--mode", "threads", "--ranks", "2", "--devices", "0,99", "--log", "10"], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "out of range" in r.stderr, r.stdout + r.stderr
