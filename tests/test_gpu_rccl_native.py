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
    _lib.check(lib.mh_marlin_test_exchange_dev(which, a.ptr, x.nbytes, b.ptr), "test_exchange_dev")
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
rc = lib.mh_marlin_test_exchange_dev(0, a.ptr, x.nbytes, b.ptr)
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
    _lib.check(lib.mh_marlin_test_exchange_dev(which, a.ptr, x.nbytes, b.ptr), "test_exchange_dev")
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
                          "--log-constraints", "14", "--no-cpu-baseline", "--no-seam-route"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["n_gpus"] == 1 and rec["value"] > 0 and rec["transport"] is None and rec["proof"]["verified"] is True
