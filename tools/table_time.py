"""Time mh_bases_precompute (the fixed-base window table) for a 3 * 2^20-point SRS (GPU box)."""
import sys, time
sys.path.insert(0, ".")
import marlin_amd as M
from marlin_amd import marlin as GM
from marlin_amd.api import Bases
M.init(0)
for n in (1 << 18, 3 << 20):
    b = Bases.srs_powers(GM.fr_mont(0x123456789abcdef), n); M.synchronize()
    t0 = time.perf_counter(); b.precompute(); M.synchronize(); t1 = time.perf_counter()
    print("n = %d: mh_bases_precompute %.1f ms, table_info %s" % (n, (t1 - t0) * 1e3, b.table_info()))
    b.free()
