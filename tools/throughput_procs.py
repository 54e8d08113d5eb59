"""Throughput of several provers sharing ONE GPU (the cheap form): P processes, each with its own context, key
and window table, prove back to back; while one waits for its host round trips (results -> Fiat-Shamir -> next round, ~0.9 ms idle
per proof) or runs a latency-bound stage (plane sums, small glue kernels), the others' kernels fill the chip.
Usage (GPU box): python tools/throughput_procs.py [log_n] [procs ...] [--pc sonic] [--secs S] [--json]
   -> one line per process count: proofs/s and constraints/s (--json: one JSON object; bench.py's `throughput_pipelined`).
Never the headline: bench.py's value is one proof at a time, as benches/bench.rs:94-107 runs them."""
import os, sys, time, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = r'''
import os, sys, time, json
sys.path.insert(0, %(root)r)
import numpy as np
import marlin_amd as M
from marlin_amd import marlin as GM
M.init(0)
n = 1 << %(log_n)d
srs = GM.universal_setup(n, n, 3 * n, 0x1f3a9c5d7e2b4a6f, 0x5eed5eed, pc=%(pc)r)
nc, ni, mats, inst, wit = GM.dummy_circuit(0x1234567, 0x7654321, 10, n)
pk = GM.index(srs, nc, ni, mats, pc=%(pc)r)
d_inst, d_wit = M.DeviceBuffer.from_numpy(np.ascontiguousarray(inst)), M.DeviceBuffer.from_numpy(np.ascontiguousarray(wit))
for _ in range(3):
    GM.prove_dev(pk, d_inst, d_wit, bytes(range(32)))
open(%(ready)r, "w").write("1")
while not os.path.exists(%(go)r):
    time.sleep(0.001)
t0 = time.time()
k = 0
while time.time() - t0 < %(secs)f:
    GM.prove_dev(pk, d_inst, d_wit, bytes(range(32)))
    k += 1
print(json.dumps({"proofs": k, "seconds": time.time() - t0}))
'''
def run(log_n, procs, secs=4.0, pc="marlin"):
    import shutil, tempfile
    d = tempfile.mkdtemp()
    go = os.path.join(d, "go")
    ps = []
    try:
        for i in range(procs):
            code = WORKER % {"root": ROOT, "log_n": log_n, "ready": os.path.join(d, "ready%d" % i), "go": go, "secs": secs, "pc": pc}
            ps.append(subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
        while not all(os.path.exists(os.path.join(d, "ready%d" % i)) for i in range(procs)):
            time.sleep(0.01)
            dead = [p for p in ps if p.poll() is not None]
            if dead:
                raise SystemExit("a worker died before it was ready (exit %d): %s" % (dead[0].returncode, dead[0].stderr.read()[-600:]))
        open(go, "w").write("1")
        outs = []
        for i, p in enumerate(ps):
            so, se = p.communicate(timeout=600)
            if p.returncode != 0 or not so.strip():
                raise SystemExit("worker %d failed (exit %d): %s" % (i, p.returncode, se[-600:]))
            outs.append(json.loads(so.strip().splitlines()[-1]))
        rate = sum(o["proofs"] / o["seconds"] for o in outs)
        return rate, outs
    finally:
        for p in ps:
            if p.poll() is None:
                p.kill()
        shutil.rmtree(d, ignore_errors=True)
if __name__ == "__main__":
    argv = sys.argv[1:]
    as_json = "--json" in argv
    pc = argv[argv.index("--pc") + 1] if "--pc" in argv else "marlin"
    secs = float(argv[argv.index("--secs") + 1]) if "--secs" in argv else 4.0
    nums = [a for i, a in enumerate(argv) if a.isdigit() and (i == 0 or argv[i - 1] not in ("--secs",))]
    log_n = int(nums[0]) if nums else 20
    plist = [int(x) for x in nums[1:]] or [1, 2, 3]
    res = {}
    for p in plist:
        rate, outs = run(log_n, p, secs, pc)
        res[p] = {"processes": p, "proofs_per_s": round(rate, 3), "constraints_per_s": round(rate * (1 << log_n), 1), "ms_per_proof_aggregate": round(1e3 / rate, 3),
                  "proofs_per_process": [o["proofs"] for o in outs], "window_s": secs}
        if not as_json:
            print("2^%d, %d process(es) on one GPU: %.2f proofs/s = %.2f M constraints/s (%.2f ms per proof aggregate; per process: %s)"
                  % (log_n, p, rate, rate * (1 << log_n) / 1e6, 1e3 / rate, ", ".join("%d in %.2fs" % (o["proofs"], o["seconds"]) for o in outs)), flush=True)
    if as_json:
        print(json.dumps(res))
