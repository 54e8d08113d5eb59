import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch.distributed as dist
import marlin_amd as M
from marlin_amd import dist as MD
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
M.init(0)
MD.enable_sharded_prove(dist)
MD.enable_alltoall(dist)
rng = np.random.default_rng(11)
def rand_fr(n):
    x = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
    x[:, 3] &= np.uint64((1 << 59) - 1)
    return x
# the distributed transforms first, as in the test
for log_n in (6, 7, 13, 20):
    n = 1 << log_n
    x = rand_fr(n)
    want = M.ntt(x)
    mine = MD.c_layout_slice(x, rank, world)
    d_in, d_out = M.DeviceBuffer.from_numpy(mine), M.DeviceBuffer(32 * (n // world))
    MD.ntt_dist_dev(d_in, d_out, log_n)
    got = d_out.download((n // world, 4))
    idx = MD.m_layout_indices(n, rank, world)
    assert np.array_equal(got, want[idx])
    owners = [None] * world
    dist.all_gather_object(owners, idx)
    MD.ntt_dist_dev(d_out, d_out, log_n, inverse=True)
    assert np.array_equal(d_out.download((n // world, 4)), mine)
    e = rand_fr(n)
    d_e = M.DeviceBuffer.from_numpy(np.ascontiguousarray(e[idx]))
    MD.ntt_dist_dev(d_e, d_e, log_n, inverse=True)
    assert np.array_equal(d_e.download((n // world, 4)), MD.c_layout_slice(M.intt(e), rank, world))
    for b in (d_in, d_out, d_e):
        b.free()
n = 1 << 15
tau = np.array([0x1234567, 0, 0, 0], dtype=np.uint64)
B = M.Bases.srs_powers(tau, n + 64)
B.precompute(14)
allb = B.download()
aff = lambda a: [tuple(M.g1_to_affine(r)[0]) for r in a]
bad = 0
for it in range(int(sys.argv[1])):
    s1, s2 = rand_fr(n), rand_fr(n - 5)
    d1, d2 = M.DeviceBuffer.from_numpy(s1), M.DeviceBuffer.from_numpy(s2)
    whole = M.msm_batch_dev([(B, 0, d1, n), (B, 37, d2, n - 5), (B, 0, d2, n - 5)])
    l1, l2 = MD.c_layout_slice(s1, rank, world), MD.c_layout_slice(s2, rank, world)
    e1, e2 = M.DeviceBuffer.from_numpy(l1), M.DeviceBuffer.from_numpy(l2)
    jobs = [(rank, e1, len(l1)), (37 + rank, e2, len(l2)), (rank, e2, len(l2))]
    got = MD.msm_batch_sliced_dev(B, jobs, world)
    if aff(got) != aff(whole):
        bad += 1
        diff = [j for j in range(3) if aff(got)[j] != aff(whole)[j]]
        part = MD.msm_batch_sliced_dev(B, jobs, world, combine=False)
        refs = []
        for j, (first, l) in enumerate(((rank, l1), (37 + rank, l2), (rank, l2))):
            Bg = M.Bases(np.ascontiguousarray(allb[first:first + world * len(l):world][:len(l)]))
            refs.append(M.msm(Bg, l)); Bg.free()
        whole2 = M.msm_batch_dev([(B, 0, d1, n), (B, 37, d2, n - 5), (B, 0, d2, n - 5)])
        print("rank", rank, "it", it, "DIFF jobs", diff, "partial(now) ok:", aff(part) == aff(refs), "whole reproducible:", aff(whole2) == aff(whole), flush=True)
    dist.barrier()
print("rank", rank, "bad", bad, flush=True)
dist.barrier(); dist.destroy_process_group()
