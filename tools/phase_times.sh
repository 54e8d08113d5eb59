#!/bin/bash
# Usage (GPU box, repo root): tools/phase_times.sh -- wall time of universal_setup / index / prove at 2^18 and 2^20
set -u
O=gpurun_out/r02m; mkdir -p $O
python - > $O/phases.txt 2>&1 <<'P'
import time, marlin_amd as M
from marlin_amd import marlin as GM
M.init(0)
for log_n in (18, 20):
    n = 1 << log_n
    t0 = time.perf_counter(); srs = GM.universal_setup(n, n, 3 * n, 0x1234567, 0x89abcd); M.synchronize(); t1 = time.perf_counter()
    ncp, ni, mats, inst, wit = GM.dummy_circuit(3, 5, 10, n); t2 = time.perf_counter()
    pk = GM.index(srs, ncp, ni, mats); M.synchronize(); t3 = time.perf_counter()
    pk2 = GM.index(srs, ncp, ni, mats); M.synchronize(); t4 = time.perf_counter()
    p = GM.prove(pk, inst, wit, bytes(32)); t5 = time.perf_counter()
    p = GM.prove(pk, inst, wit, bytes(32)); t6 = time.perf_counter()
    print("2^%d: universal_setup %.3f s, circuit build (python) %.3f s, index (first, builds window table) %.3f s, index (again) %.3f s, prove first %.3f s, prove %.3f s"
          % (log_n, t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5))
    pk.free(); pk2.free()
P
cat $O/phases.txt
