import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import marlin_amd as M
from marlin_amd import dist as MD
M.init(0)
tag = sys.argv[1]
rng = np.random.default_rng(11)
def rand_fr(n):
    x = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
    x[:, 3] &= np.uint64((1 << 59) - 1)
    return x
aff = lambda a: [tuple(M.g1_to_affine(r)[0]) for r in a]
n, c = 1 << 15, 14
tau = np.array([0x1234567, 0, 0, 0], dtype=np.uint64)
B = M.Bases.srs_powers(tau, n + 64)
B.precompute(c)
allb = B.download()
bad = 0
for trial in range(6):
    s1, s2 = rand_fr(n), rand_fr(n - 5)
    d1, d2 = M.DeviceBuffer.from_numpy(s1), M.DeviceBuffer.from_numpy(s2)
    whole = M.msm_batch_dev([(B, 0, d1, n), (B, 37, d2, n - 5), (B, 0, d2, n - 5)])
    whole2 = M.msm_batch_dev([(B, 0, d1, n), (B, 37, d2, n - 5), (B, 0, d2, n - 5)])
    if aff(whole) != aff(whole2):
        print(tag, "WHOLE not reproducible", trial); bad += 1
    world = 8
    parts = []
    for rank in range(world):
        l1, l2 = MD.c_layout_slice(s1, rank, world), MD.c_layout_slice(s2, rank, world)
        e1, e2 = M.DeviceBuffer.from_numpy(l1), M.DeviceBuffer.from_numpy(l2)
        got = MD.msm_batch_sliced_dev(B, [(rank, e1, len(l1)), (37 + rank, e2, len(l2)), (rank, e2, len(l2))], world, combine=False)
        got2 = MD.msm_batch_sliced_dev(B, [(rank, e1, len(l1)), (37 + rank, e2, len(l2)), (rank, e2, len(l2))], world, combine=False)
        if aff(got) != aff(got2):
            print(tag, "PARTIAL not reproducible", trial, rank); bad += 1
        parts.append(got)
    tot = [MD.g1_sum(np.stack([p[j] for p in parts])) for j in range(3)]
    if aff(tot) != aff(whole):
        print(tag, "DIFFERS", trial); bad += 1
print(tag, "done bad =", bad)
