"""MH_TRACE=1 python tools/trace_prove.py [log_n]: phase timing of one Marlin::prove (reference's print-trace labels)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import marlin_amd as M
from marlin_amd import marlin as GM
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
M.init(0)
n = 1 << log_n
t0 = time.time()
srs = GM.universal_setup(n, n, 3 * n, 0x1f3a9c5d7e2b4a6f, 0x5eed5eed)
print("universal_setup %.2fs" % (time.time() - t0)); t0 = time.time()
nc, ni, mats, inst, wit = GM.dummy_circuit(0x1234567, 0x7654321, 10, n)
pk = GM.index(srs, nc, ni, mats)
print("index %.2fs" % (time.time() - t0))
GM.prove(pk, inst, wit, bytes(32))
os.environ["MH_TRACE"] = "1"
t0 = time.time()
GM.prove(pk, inst, wit, bytes(32))
print("prove %.1f ms" % ((time.time() - t0) * 1e3))
