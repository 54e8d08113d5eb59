"""Per-rank cost of the distributed transform (mh_ntt_dist_dev) measured on ONE GPU: this process plays rank R of G with the
all-to-all replaced by a local copy (marlin_amd.dist.enable_simulated_alltoall), so the kernels a rank runs -- the local
m-point transform, the twiddle multiplication, the G-point transforms -- are timed as they are; the exchange is reported as
bytes (32 n / G^2 to each peer) and priced at xGMI's ~50 GB/s per link in one direction for a many-to-many pattern.
    python tools/ntt_dist_bench.py [G=8] [log_n ...]"""
import sys, time
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import marlin_amd as M
from marlin_amd import dist as MD, _lib

G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
logs = [int(a) for a in sys.argv[2:]] or [20, 22, 23, 24, 25]
M.init(0)
L = _lib.load()
for lg in logs:
    n = 1 << lg
    full = M.DeviceBuffer(32 * n)
    full.upload(np.random.default_rng(lg).integers(0, 1 << 60, size=(n, 4), dtype=np.uint64))
    out = M.DeviceBuffer(32 * n)
    res = {}
    for mode in ("single", "rank"):
        if mode == "rank":
            MD.enable_simulated_alltoall(3 % G, G)
        for inverse in (False, True):
            f = (lambda: M.ntt_dev(full, out, lg, inverse=inverse)) if mode == "single" else (lambda: MD.ntt_dist_dev(full, out, lg, inverse=inverse))
            f(); f(); M.synchronize()
            reps = 10
            t0 = time.perf_counter()
            for _ in range(reps):
                f()
            M.synchronize()
            res[(mode, inverse)] = (time.perf_counter() - t0) * 1e3 / reps
        MD.disable_sharded_prove()
    peer_bytes = 32 * n // (G * G)
    a2a_ms = peer_bytes / 50e9 * 1e3
    print("2^%d over G=%d: one GPU %.3f / %.3f ms (fwd / inv); one rank's kernels + local copy %.3f / %.3f ms; exchange %d B to each of %d peers "
          "(~%.3f ms at 50 GB/s per link) -> ~%.3f ms per rank, %.1fx" % (
              lg, G, res[("single", False)], res[("single", True)], res[("rank", False)], res[("rank", True)], peer_bytes, G - 1, a2a_ms,
              res[("rank", False)] + a2a_ms, res[("single", False)] / (res[("rank", False)] + a2a_ms)))
    full.free(); out.free()
