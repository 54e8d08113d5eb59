"""The native transport: RCCL called by the library itself on its own stream (marlin_amd/csrc/rccl_native.h; replaces the
Python callbacks of marlin_amd/dist.py in the exchange path of a sharded Marlin::prove -- /root/reference src/lib.rs:172,193,213
are the commitments whose MSMs are sharded, src/ahp/prover.rs:532-535,655-688 the transforms that are distributed).

This box has ONE GPU and RCCL refuses two ranks on one device (and the card refuses CPX partitioning,
profiles/r04a_cpx_partition_attempt.txt), so what can run here is a communicator of one rank: ncclCommInitRank, ncclAllGather
and ncclAllToAll execute for real on the library's stream, against the library's buffers, inside and outside a proof.  The
N > 1 logic around them (payload layout, share / whole flags, slices) is the transport-independent part that the gloo tests of
tests/test_gpu_marlin.py run at 2 / 3 / 4 / 8 ranks with the same `allgather2` / device all-gather code path."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

NATIVE_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np
%(import_torch)s
import marlin_amd as M
from marlin_amd import dist as MD, marlin as GM, _lib
lib = _lib.load()
M.init(0)
n = 1 << 12
srs = GM.universal_setup(n, n, 3 * n, 0x1234567, 0x7654321)
ncp, ni, mats, inst, wit = GM.dummy_circuit(3, 5, 10, n)
pk = GM.index(srs, ncp, ni, mats)
want = GM.prove(pk, inst, wit, bytes(range(32)))
assert MD.native_rccl_info()["active"] is False
assert MD.enable_native_rccl(None), "native transport refused: " + (lib.mh_last_error() or b"").decode()
info = MD.native_rccl_info()
assert info["active"] and "librccl" in info["librccl"], info
# the three collectives, by themselves: host all-gather (the partial points of a commit round) ...
for _ in range(3):
    assert MD.selftest_allgather(None)
# ... all-to-all and device all-gather on device buffers, stream-ordered on the library's stream
x = np.random.default_rng(5).integers(0, 1 << 62, size=(1 << 14, 4), dtype=np.uint64)
a, b = M.DeviceBuffer.from_numpy(x), M.DeviceBuffer(x.nbytes)
for which in (0, 1, 0, 1):
    b.upload(np.zeros_like(x))
    _lib.check(lib.mh_marlin_probe_exchange_dev(which, a.ptr, x.nbytes, b.ptr), "test_exchange_dev")
    assert np.array_equal(b.download(x.shape), x), which
info = MD.native_rccl_info()
assert info["allgather_host"] == 3 and info["alltoall"] == 2 and info["allgather_dev"] == 2, info
calls, host_ms = MD.exchange_stats()
assert calls == 4 and host_ms > 0
# a proof with the communicator alive is the same proof (world = 1: nothing is sharded, nothing may change)
assert GM.prove(pk, inst, wit, bytes(range(32))) == want
# a callback transport replaces the native one, and the native one can come back
MD.disable_sharded_prove()
assert MD.native_rccl_info()["active"] is False
assert MD.enable_native_rccl(None, sliced=False)
assert MD.selftest_allgather(None)
rc = lib.mh_marlin_probe_exchange_dev(0, a.ptr, x.nbytes, b.ptr)
assert rc != 0, "sliced = False must unregister the all-to-all"
MD.disable_sharded_prove()
M.shutdown() if hasattr(M, "shutdown") else lib.mh_shutdown()
print("native rccl world=1 ok:", info["librccl"])
'''


@pytest.mark.parametrize("with_torch", [True, False], ids=["torch-process", "plain-process"])
def test_native_rccl_world_1(gpu, tmp_path, with_torch):
    """mh_rccl_unique_id -> mh_marlin_set_rccl(0, 1, id) -> the registered collectives run through RCCL on the library's stream.
    Twice: in a process that has imported torch (librccl.so.1 is then the copy torch mapped -- the situation of bench.py) and
    in one that has not (the loader finds /opt/rocm/lib's -- the situation of a C or Rust caller)."""
    script = tmp_path / "native_worker.py"
    script.write_text(NATIVE_WORKER % {"root": ROOT, "import_torch": "import torch" if with_torch else "assert 'torch' not in sys.modules"})
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "native rccl world=1 ok" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
    if with_torch:
        assert "torch" in r.stdout.split("ok:")[-1], r.stdout      # the copy of librccl torch ships, not a second one


def test_device_allgather_callback_over_rccl_world_1(gpu, tmp_path):
    """The callback transport's device all-gather (marlin_amd/dist.py: all_gather_into_tensor on zero-copy views of the library's
    buffers, stream-ordered) over RCCL with one rank -- the fallback bench.py takes when the native transport fails its self-test."""
    code = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
import marlin_amd as M
from marlin_amd import dist as MD, _lib
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
M.init(0)
lib = _lib.load()
MD.enable_sharded_prove(dist, device=torch.device("cuda", 0))
ts = MD.use_torch_stream(torch.device("cuda", 0))
MD.enable_alltoall(dist, device=torch.device("cuda", 0), stream=ts)
x = np.random.default_rng(6).integers(0, 1 << 62, size=(1 << 13, 4), dtype=np.uint64)
a, b = M.DeviceBuffer.from_numpy(x), M.DeviceBuffer(x.nbytes)
for which in (1, 0, 1):
    b.upload(np.zeros_like(x))
    _lib.check(lib.mh_marlin_probe_exchange_dev(which, a.ptr, x.nbytes, b.ptr), "test_exchange_dev")
    assert np.array_equal(b.download(x.shape), x), which
st = MD._keepalive["a2a_state"]
assert st["zero_copy"] and st["stream_ordered"] and st["calls"] == 3, st
print("device all-gather over rccl ok")
dist.destroy_process_group()
''' % {"root": ROOT}
    script = tmp_path / "ag_worker.py"
    script.write_text(code)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29793", WORLD_SIZE="1", RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "device all-gather over rccl ok" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


def test_bench_world_1_under_torchrun_uses_no_transport(gpu):
    """`torchrun --nproc-per-node 1 bench.py --gpus 1` (how the driver launches N ranks, at N = 1): the line is the one-GPU line."""
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                          "--master-port", "29831", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1",
                          "--log-constraints", "14", "--no-cpu-baseline", "--no-throughput", "--no-seam-route"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["n_gpus"] == 1 and rec["value"] > 0 and rec["transport"] is None and rec["proof"]["verified"] is True


# ---- N > 1 through the native transport, on ONE GPU: the shared-memory stand-in for librccl (tests/mock_rccl/) ----------------
MOCK = os.path.join(ROOT, "tests", "mock_rccl", "libmock_rccl.so")
MOCK_NOA2A = os.path.join(ROOT, "tests", "mock_rccl", "libmock_rccl_noa2a.so")      # the same without ncclAllToAll: grouped ncclSend / ncclRecv

MOCK_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import torch.distributed as dist
import marlin_amd as M
from marlin_amd import marlin as GM, dist as MD
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)      # used ONCE: to hand rank 0's unique id to the others
M.init(0)                                                          # every rank on the box's one GPU
n = 1 << %(log_n)d
srs = GM.universal_setup(n, n, 3 * n, %(tau)d, %(gamma)d, pc=%(pc)r)
nc, ni, mats, inst, wit = GM.dummy_circuit(%(a)d, %(b)d, 10, n)
pk = GM.index(srs, nc, ni, mats, pc=%(pc)r)
assert MD.enable_native_rccl(dist, sliced=bool(%(sliced)d))
info = MD.native_rccl_info()
assert info["active"] and "mock_rccl" in info["librccl"], info
assert MD.selftest_allgather(dist)
if %(sliced)d and world & (world - 1) == 0:
    assert MD.selftest_alltoall(dist, log_n=10)
before = MD.native_rccl_info()
proof = GM.prove(pk, inst, wit, bytes(range(32)))
proof2 = GM.prove(pk, inst, wit, bytes(range(1, 33)))
after = MD.native_rccl_info()
open(os.path.join(%(out)r, "proof%%d.bin" %% rank), "wb").write(proof + proof2)
open(os.path.join(%(out)r, "info%%d.txt" %% rank), "w").write("%%d %%d %%d" %% tuple(after[k] - before[k] for k in ("allgather_host", "alltoall", "allgather_dev")))
dist.barrier(); MD.disable_sharded_prove(); dist.destroy_process_group()
'''


@pytest.mark.parametrize("world,log_n,pc,sliced,variant", [(2, 12, "marlin", 0, "sync"), (4, 16, "marlin", 0, "sync"), (2, 12, "marlin", 1, "sync"),
                                                           (4, 12, "sonic", 1, "sync"), (4, 16, "marlin", 1, "sync"), (8, 16, "marlin", 1, "sync"),
                                                           (8, 16, "sonic", 1, "sync"), (3, 12, "marlin", 1, "sync"),
                                                           (4, 14, "marlin", 1, "async"), (8, 16, "marlin", 1, "async"), (4, 12, "sonic", 1, "async-noa2a"),
                                                           (2, 12, "marlin", 0, "async")])
def test_native_transport_at_n_ranks_gives_the_single_gpu_proof(gpu, tmp_path, world, log_n, pc, sliced, variant):
    """mh_marlin_set_rccl with N = 2 / 3 / 4 / 8 ranks -- the C++ all-gather of partial points, the all-to-all of the distributed
    transforms and the device all-gather of round polynomials (rccl_native.h), entered from mh_marlin_prove exactly as on a
    multi-GPU node -- with the collectives carried by the shared-memory stand-in for librccl (RCCL refuses two ranks on the one
    device of this box): every rank's two proofs are the unsharded prover's bytes, and the counters show which collectives ran
    (sliced = 1 and a power-of-two world: distributed transforms and two device all-gathers per proof).
    variant "async" (ADVICE r04): the stand-in ENQUEUES its collectives -- asynchronous copies and the barriers as host functions in
    stream order, the call returns at once, like RCCL's -- so the library's reuse of sl_send / sl_recv / ntt_dist_buf and the host
    all-gather's staging are exercised the way a real communicator exercises them; "noa2a": the stand-in has no ncclAllToAll and
    the transport takes its grouped ncclSend / ncclRecv path."""
    from marlin_amd import marlin as GM
    import json
    GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "marlin_proofs.json")))
    TAU, GAMMA = int(GOLD["tau"], 16), int(GOLD["gamma"], 16)
    assert os.path.exists(MOCK), "tests/mock_rccl/libmock_rccl.so not built (python -c 'import __graft_entry__ as g; g.build()')"
    a, b = 0x1234567, 0x7654321
    n = 1 << log_n
    srs = GM.universal_setup(n, n, 3 * n, TAU, GAMMA, pc=pc)
    nc, ni, mats, inst, wit = GM.dummy_circuit(a, b, 10, n)
    pk = GM.index(srs, nc, ni, mats, pc=pc)
    want = GM.prove(pk, inst, wit, bytes(range(32))) + GM.prove(pk, inst, wit, bytes(range(1, 33)))
    script = tmp_path / "mock_worker.py"
    script.write_text(MOCK_WORKER % {"root": ROOT, "out": str(tmp_path), "tau": TAU, "gamma": GAMMA, "a": a, "b": b, "log_n": log_n, "pc": pc, "sliced": sliced})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29913 + world + log_n + 40 * sliced + (7 if pc == "sonic" else 0) + 100 * len(variant)),
               WORLD_SIZE=str(world), MH_RCCL_LIB=MOCK_NOA2A if "noa2a" in variant else MOCK, MH_CHECK="1")
    if "async" in variant:
        env["MH_MOCK_RCCL_ASYNC"] = "1"
    if sliced:
        env["MH_SLICED"] = "2"               # also with 2 ranks, where the library would keep the rounds replicated
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, so[-800:] + se[-2500:]
    pow2 = world & (world - 1) == 0
    for r in range(world):
        assert open(tmp_path / ("proof%d.bin" % r), "rb").read() == want, r
        ag_host, a2a, ag_dev = (int(x) for x in open(tmp_path / ("info%d.txt" % r)).read().split())
        assert ag_host >= 8                                        # >= 4 per proof: one per commit round and the openings
        if sliced and pow2 and n >= world * world:
            assert a2a >= 2 * 9 and ag_dev == 2 * 2, (a2a, ag_dev)   # nine distributed transforms, two round gathers per proof
        else:
            assert a2a == 0 and ag_dev == 0


@pytest.mark.parametrize("world,log_n", [(4, 14), (8, 16), (2, 14)])
def test_bench_native_transport_with_n_ranks(gpu, world, log_n):
    """`python bench.py --gpus N --transport native` (N = 2, 4, 8) on this box's one GPU, the collectives carried by the stand-in:
    the set-up the driver's 8-GPU run goes through -- unique id handed over, all-gather and distributed-transform self-tests
    agreed on by all ranks, sliced rounds from 4 ranks on -- and a line that explains itself (transport, per-rank breakdown with
    the exchanges, every rank holding the one-GPU proof)."""
    import json
    assert os.path.exists(MOCK)
    env = dict(os.environ, BENCH_BACKEND="gloo", BENCH_SINGLE_DEVICE="1", MH_RCCL_LIB=MOCK)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--rehearsal", "--steps", "2", "--warmup", "1", "--transport", "native",
                          "--log-constraints", str(log_n), "--no-cpu-baseline", "--no-throughput"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["n_gpus"] == world and rec["value"] > 0 and len(rec["per_rank"]) == world
    assert rec["transport"]["kind"] == "native-rccl" and rec["proof"]["verified"] is True and rec["proof"]["identical_on_all_ranks"] is True, (rec["transport"], rec["proof"])
    if world < 4:                                         # 2 ranks: the rounds stay replicated (the exchanges would cost more than they save)
        assert "replicated" in rec["config"]["parallelism"] and rec["transport"]["native_rccl"]["alltoall"] == 0
        return
    assert "slices" in rec["config"]["parallelism"], rec["config"]
    assert rec["transport"]["kind"] == "native-rccl" and rec["transport"]["native_rccl"]["alltoall"] > 0, rec["transport"]
    assert rec["proof"]["verified"] is True and rec["proof"]["identical_on_all_ranks"] is True, rec["proof"]
    for r in rec["per_rank"]:
        b = r["breakdown_ms_per_step"]
        assert b["exchanges_per_step"] >= 11 and b["exchange_on_stream"] > 0 and b["exchange_host_wall"] > 0, r


def test_bench_falls_back_to_the_callbacks_when_one_rank_fails_the_native_self_test(gpu):
    """The stand-in damages what rank 1 receives: the native all-gather self-test fails THERE only, the ranks agree on the outcome
    (all-reduce MIN) before anyone enters another collective, every rank tears the native transport down and the run proceeds on
    the torch.distributed callbacks -- a line with the right proof instead of a hang or a wrong commitment."""
    import json
    env = dict(os.environ, BENCH_BACKEND="gloo", BENCH_SINGLE_DEVICE="1", MH_RCCL_LIB=MOCK, MH_MOCK_RCCL_CORRUPT_RANK="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--rehearsal", "--steps", "1", "--warmup", "1", "--transport", "native",
                          "--log-constraints", "14", "--no-cpu-baseline", "--no-throughput"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "native RCCL transport unavailable" in out.stderr, out.stderr[-4000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["transport"]["kind"] == "callback-torch.distributed-gloo" and rec["transport"]["native_rccl"] is None, rec["transport"]
    assert rec["proof"]["verified"] is True and rec["proof"]["identical_on_all_ranks"] is True and "slices" in rec["config"]["parallelism"]


FAIL_WORKER = r'''
import os, sys, time
sys.path.insert(0, %(root)r)
import ctypes as C
import torch.distributed as dist
import marlin_amd as M
from marlin_amd import marlin as GM, dist as MD, _lib
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
M.init(0)
lib = _lib.load()
n = 1 << %(log_n)d
srs = GM.universal_setup(n, n, 3 * n, %(tau)d, %(gamma)d)
nc, ni, mats, inst, wit = GM.dummy_circuit(%(a)d, %(b)d, 10, n)
pk = GM.index(srs, nc, ni, mats)
assert MD.enable_native_rccl(dist, sliced=bool(%(sliced)d))
seed = bytes(range(32))
first = GM.prove(pk, inst, wit, seed)                     # warm: every buffer has its size now
c0, c1 = C.c_uint64(), C.c_uint64()
_lib.check(lib.mh_debug_fail_scratch(0, C.byref(c0)), "hook")
assert GM.prove(pk, inst, wit, seed) == first
_lib.check(lib.mh_debug_fail_scratch(0, C.byref(c1)), "hook")
per_proof = c1.value - c0.value                            # scratch requests of one proof on this rank
assert per_proof > 20, per_proof
log = []
for frac in %(fracs)r:
    dist.barrier()
    if rank == %(victim)d:
        _lib.check(lib.mh_debug_fail_scratch(int(per_proof * frac) + 1, None), "hook")
    t0 = time.time()
    try:
        GM.prove(pk, inst, wit, seed)
        outcome = "ok"
    except _lib.MarlinHipError as e:
        outcome = "failed: " + str(e)[:160].replace("\n", " ")
    dt = time.time() - t0
    _lib.check(lib.mh_debug_fail_scratch(0, None), "hook")
    again = GM.prove(pk, inst, wit, seed) == first         # the transport is still in step: the next proof is the right proof
    log.append("%%.2f %%.3f %%s %%s" %% (frac, dt, again, outcome))
open(os.path.join(%(out)r, "fail%%d.txt" %% rank), "w").write("\n".join(log))
open(os.path.join(%(out)r, "first%%d.bin" %% rank), "wb").write(first)
dist.barrier(); MD.disable_sharded_prove(); dist.destroy_process_group()
'''


@pytest.mark.parametrize("world,sliced,victim,enqueued", [(4, 1, 2, False), (2, 0, 1, False), (4, 1, 0, True)])
def test_a_rank_that_fails_mid_prove_fails_the_job_on_every_rank(gpu, tmp_path, world, sliced, victim, enqueued):
    """a rank that fails locally inside a sharded proof must fail the JOB, not hang it.  One rank's request for
    device scratch is forced to fail (mh_debug_fail_scratch) a quarter, half and most of the way through a proof -- inside the sliced
    rounds with their all-to-alls and round gathers at 4 ranks, inside the replicated rounds at 2: the poisoned rank keeps entering
    the collectives up to the next all-gather of partial points, whose error word makes EVERY rank return non-zero from the same
    commit round, within seconds; and because all ranks left the proof at the same collective, the very next proof is again the
    one-GPU proof on every rank."""
    from marlin_amd import marlin as GM
    import json
    GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "marlin_proofs.json")))
    TAU, GAMMA = int(GOLD["tau"], 16), int(GOLD["gamma"], 16)
    assert os.path.exists(MOCK)
    a, b, log_n = 0x1234567, 0x7654321, 12
    n = 1 << log_n
    srs = GM.universal_setup(n, n, 3 * n, TAU, GAMMA)
    nc, ni, mats, inst, wit = GM.dummy_circuit(a, b, 10, n)
    pk = GM.index(srs, nc, ni, mats)
    want = GM.prove(pk, inst, wit, bytes(range(32)))
    fracs = (0.22, 0.5, 0.8)
    script = tmp_path / "fail_worker.py"
    script.write_text(FAIL_WORKER % {"root": ROOT, "out": str(tmp_path), "tau": TAU, "gamma": GAMMA, "a": a, "b": b, "log_n": log_n, "sliced": sliced,
                                     "victim": victim, "fracs": fracs})
    from tests.util import hooks_env          # mh_debug_fail_scratch lives in the hooks library (the product's objects + testhooks.hip)
    env = hooks_env(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29871 + world + 10 * enqueued), WORLD_SIZE=str(world), MH_RCCL_LIB=MOCK, MH_CHECK="1")
    if sliced:
        env["MH_SLICED"] = "2"
    if enqueued:
        env["MH_MOCK_RCCL_ASYNC"] = "1"          # collectives enqueued in stream order, as RCCL's are
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, so[-800:] + se[-2500:]
    for r in range(world):
        assert open(tmp_path / ("first%d.bin" % r), "rb").read() == want
        lines = open(tmp_path / ("fail%d.txt" % r)).read().splitlines()
        assert len(lines) == len(fracs)
        for line in lines:
            frac, dt, again, outcome = line.split(" ", 3)
            assert outcome.startswith("failed"), (r, line)          # non-zero on EVERY rank, the victim included
            assert float(dt) < 10.0, (r, line)
            assert again == "True", (r, line)
        if r == victim:
            assert all("forced by mh_debug_fail_scratch" in l for l in lines), lines     # the victim reports its own cause
