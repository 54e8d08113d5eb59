"""Times single transforms of 2^lg points on device buffers (HIP events through mh_prof_*):
python tools/ntt_bench.py [lg ...]  -> ms per transform, G butterflies/s, GB/s of the algorithmic 64 B per point."""
import sys
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import marlin_amd as M
from marlin_amd import _lib
import ctypes as C

M.init(0)
L = _lib.load()
for lg in [int(a) for a in sys.argv[1:]] or [16, 18, 20, 22, 23]:
    n = 1 << lg
    a, b = C.c_void_p(), C.c_void_p()
    _lib.check(L.mh_alloc(32 * n, C.byref(a)), "alloc"); _lib.check(L.mh_alloc(32 * n, C.byref(b)), "alloc")
    host = np.random.default_rng(lg).integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
    host[:, 3] &= (1 << 60) - 1
    _lib.check(L.mh_memcpy_h2d(a, host.ctypes.data, 32 * n), "h2d")
    for inverse in (0, 1):
        for _ in range(2):
            _lib.check(L.mh_ntt_dev(0 if _lib.CURVE == "bls12_381" else 1, a, b, lg, inverse), "ntt")
        M.prof_enable(True); M.prof_reset()
        reps = 10
        for _ in range(reps):
            _lib.check(L.mh_ntt_dev(0 if _lib.CURVE == "bls12_381" else 1, a, b, lg, inverse), "ntt")
        M.synchronize()
        ms, launches = M.prof_get(0)
        ms /= reps
        print("2^%d %s: %.4f ms  %.1f G butterflies/s  %.0f GB/s (64 B/pt)  passes=%d" % (
            lg, "inv" if inverse else "fwd", ms, n / 2 * lg / ms / 1e6, 64.0 * n / ms / 1e6, launches // reps))
    L.mh_free(a); L.mh_free(b)
