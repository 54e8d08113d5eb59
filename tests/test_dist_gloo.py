"""world_size-2 test (gloo, CPU) of the multi-GPU MSM path: shard ranges, all_gather of partial
points and the host combine (marlin_amd/dist.py + mh_g1_sum).  The per-rank partial MSMs, which run
on the GPU in production, are produced here by the oracle's C restatement (test infrastructure)."""
import os
import subprocess
import sys
import numpy as np
from oracle import cref, curve as EC, fields as F
from marlin_amd import dist as D
from tests.util import jac_np_to_affine, rand_fr, fr_to_np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np
import torch.distributed as dist
from oracle import cref
from marlin_amd import dist as D
from tests.util import rand_fr, fr_to_np
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
sizes = [1000, 1, 777]
bases, _ = cref.bases_arith(1000)
partials = []
for j, n in enumerate(sizes):
    sc = fr_to_np(rand_fr(n, 100 + j))
    lo, hi = D.shard_range(n, rank, world)
    partials.append(cref.msm(bases[lo:hi], sc[lo:hi]))
total = D.combine_partials(D.allgather_partials(np.stack(partials), dist))
np.save(os.path.join(%(out)r, "rank%%d.npy" %% rank), total)
dist.barrier()
dist.destroy_process_group()
'''


WORKER_CB = r'''
import os, sys, ctypes as C
sys.path.insert(0, %(root)r)
import numpy as np
import torch.distributed as dist
from marlin_amd import dist as D, _lib
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
D.enable_sharded_prove(dist)
send = np.arange(18, dtype=np.uint64) + 1000 * rank
recv = np.zeros(18 * world, dtype=np.uint64)
rc = _lib.load().mh_marlin_probe_allgather(send.ctypes.data, send.nbytes, recv.ctypes.data)
assert rc == 0
for g in range(world):
    assert (recv[18 * g: 18 * (g + 1)] == np.arange(18, dtype=np.uint64) + 1000 * g).all()
# payloads of different sizes alternate inside a proof (4, 4, 3, 2 jobs per exchange): the staging buffers only grow and every
# call works on views of the size it needs
for words in (74, 38, 74, 20, 600, 38, 74):
    send = np.arange(words, dtype=np.uint64) * 3 + 1000 * rank
    recv = np.zeros(words * world, dtype=np.uint64)
    assert _lib.load().mh_marlin_probe_allgather(send.ctypes.data, send.nbytes, recv.ctypes.data) == 0
    for g in range(world):
        assert (recv[words * g: words * (g + 1)] == np.arange(words, dtype=np.uint64) * 3 + 1000 * g).all()
D.disable_sharded_prove()
dist.barrier(); dist.destroy_process_group()
'''


def test_sharded_prove_allgather_callback_gloo(tmp_path):
    """the library-side exchange hook of the sharded prover (mh_marlin_set_shard) driven by gloo."""
    script = tmp_path / "worker_cb.py"
    script.write_text(WORKER_CB % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29612", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r))) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=240) == 0


def test_shard_ranges_cover_and_balance():
    for n in [0, 1, 7, 1000, 2 ** 22 - 1]:
        for world in [1, 2, 3, 8]:
            r = [D.shard_range(n, g, world) for g in range(world)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[i][1] == r[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in r]
            assert max(sizes) - min(sizes) <= 1


def test_g1_sum_matches_oracle():
    pts = [cref.g1_mul_gen(k) for k in [5, 7, F.R_MOD - 12, 0, 12]]
    s = D.g1_sum(np.stack(pts))
    assert jac_np_to_affine(s) == EC.scalar_mul(EC.G1_GEN, (5 + 7 - 12 + 12) % F.R_MOD)
    assert jac_np_to_affine(D.g1_sum(np.stack([cref.g1_mul_gen(3), cref.g1_mul_gen(F.R_MOD - 3)]))) is None


def test_sharded_msm_two_ranks_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT, "out": str(tmp_path)})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29611", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r))) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=240) == 0
    r0, r1 = np.load(tmp_path / "rank0.npy"), np.load(tmp_path / "rank1.npy")
    bases, dl = cref.bases_arith(1000)
    for j, n in enumerate([1000, 1, 777]):
        sc = rand_fr(n, 100 + j)
        want = EC.scalar_mul(EC.G1_GEN, sum(s * a for s, a in zip(sc, dl)) % F.R_MOD)
        assert jac_np_to_affine(r0[j]) == want
        assert jac_np_to_affine(r1[j]) == want          # every rank derives the same commitment


def test_bench_gpus_flag_launches_that_many_ranks():
    """`python bench.py --gpus 2` (the plain command shape, no launcher) must become 2 ranks: it re-executes itself under
    torch.distributed.run.  BENCH_DRY_RUN=1 keeps the rank plumbing (rendezvous on 127.0.0.1, barrier, MAX all_reduce,
    one JSON line from rank 0) and skips the device work, which this CPU box cannot do."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BENCH_DRY_RUN="1", BENCH_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout                      # rank 0 only
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["dry_run"] is True and rec["max_over_ranks"] == 2.0
    # launched by a launcher (RANK / WORLD_SIZE present) it must NOT spawn again: world comes from the environment
    env1 = dict(env, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1"], env=env1, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])["n_gpus"] == 1
