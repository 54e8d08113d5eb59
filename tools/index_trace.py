import time, marlin_amd as M
from marlin_amd import marlin as GM
M.init(0)
n = 1 << 20
srs = GM.universal_setup(n, n, 3 * n, 0x1234567, 0x89abcd)
ncp, ni, mats, inst, wit = GM.dummy_circuit(3, 5, 10, n)
pk = GM.index(srs, ncp, ni, mats); pk.free()
t=time.perf_counter(); pk = GM.index(srs, ncp, ni, mats); M.synchronize(); print("index again %.1f ms" % ((time.perf_counter()-t)*1e3))
