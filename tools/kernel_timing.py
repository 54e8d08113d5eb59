"""Quick device-resident timing of the NTT and MSM kernels (development tool)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import marlin_amd as M

M.init(0)
print(M.device_info())
rng = np.random.default_rng(0)

def rand_fr(n):
    x = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
    x[:, 3] &= np.uint64((1 << 61) - 1)
    return x

M.prof_enable(True)
for log_n in [16, 18, 20, 21, 22, 23, 24]:
    n = 1 << log_n
    x = rand_fr(n)
    a = M.DeviceBuffer.from_numpy(x); b = M.DeviceBuffer(x.nbytes)
    M.ntt_dev(a, b, log_n); M.synchronize()
    M.prof_reset()
    reps = 5
    for _ in range(reps):
        M.ntt_dev(a, b, log_n)
    ms, k = M.prof_get(0)
    ms /= reps
    print("NTT 2^%d: %.3f ms  algorithmic %.1f GB/s  (%.2f Gbutterfly/s)" % (log_n, ms, 64.0 * n / ms / 1e6, n / 2 * log_n / ms / 1e6))
    a.free(); b.free()

# MSM with synthetic bases: reuse a small set of valid points tiled (distinct enough for timing)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tests.util import arith_bases, points_to_np
pts, _ = arith_bases(4096)
pnp = points_to_np(pts)
for log_n in [12, 16, 18, 20, 22]:
    n = 1 << log_n
    reps_tile = n // 4096
    big = np.tile(pnp, (reps_tile, 1))
    B = M.Bases(big)
    s = M.DeviceBuffer.from_numpy(rand_fr(n))
    M.msm_dev(B, s, n); M.synchronize()
    M.prof_reset()
    t0 = time.time()
    reps = 3
    for _ in range(reps):
        M.msm_dev(B, s, n)
    wall = (time.time() - t0) / reps * 1e3
    ms, _ = M.prof_get(1); acc, _ = M.prof_get(2)
    print("MSM 2^%d: kernels %.3f ms (accum %.3f ms) wall %.3f ms  %.2f Mpts/s  algorithmic %.1f GB/s" % (
        log_n, ms / reps, acc / reps, wall, n / (ms / reps) / 1e3, 128.0 * n / (ms / reps) / 1e6))
    B.free(); s.free()
