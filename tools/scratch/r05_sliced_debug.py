import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import marlin_amd as M
from marlin_amd import dist as MD
M.init(0)
rng = np.random.default_rng(11)
def rand_fr(n):
    x = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
    x[:, 3] &= np.uint64((1 << 59) - 1)
    return x
aff = lambda a: [tuple(M.g1_to_affine(r)[0]) for r in a]
for (msm_log, c) in ((15, 14), (13, 14), (12, 12), (15, 16)):
    n = 1 << msm_log
    tau = np.array([0x1234567, 0, 0, 0], dtype=np.uint64)
    B = M.Bases.srs_powers(tau, n + 64)
    B.precompute(c)
    for trial in range(3):
        s1, s2 = rand_fr(n), rand_fr(n - 5)
        d1, d2 = M.DeviceBuffer.from_numpy(s1), M.DeviceBuffer.from_numpy(s2)
        whole = M.msm_batch_dev([(B, 0, d1, n), (B, 37, d2, n - 5), (B, 0, d2, n - 5)])
        for world in (2, 4, 8, 16):
            parts = []
            for rank in range(world):
                l1, l2 = MD.c_layout_slice(s1, rank, world), MD.c_layout_slice(s2, rank, world)
                e1, e2 = M.DeviceBuffer.from_numpy(l1), M.DeviceBuffer.from_numpy(l2)
                got = MD.msm_batch_sliced_dev(B, [(rank, e1, len(l1)), (37 + rank, e2, len(l2)), (rank, e2, len(l2))], world, combine=False)
                parts.append(got)
                # each partial also against the variable-base path on gathered bases
                allb = B.download()
                for j, (first, l) in enumerate(((rank, l1), (37 + rank, l2), (rank, l2))):
                    Bg = M.Bases(np.ascontiguousarray(allb[first:first + world * len(l):world][:len(l)]))
                    os.environ["X"] = "1"
                    ref = M.msm(Bg, l)
                    if tuple(M.g1_to_affine(got[j])[0]) != tuple(M.g1_to_affine(ref)[0]):
                        print("MISMATCH partial", msm_log, c, trial, world, rank, j, len(l))
                    Bg.free()
            tot = [MD.g1_sum(np.stack([p[j] for p in parts])) for j in range(3)]
            print(msm_log, c, trial, world, "ok" if aff(tot) == aff(whole) else "DIFFERS")
    B.free()
