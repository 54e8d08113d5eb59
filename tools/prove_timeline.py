"""Dispatch-by-dispatch timeline of ONE prove in a rocprofv3 --kernel-trace CSV of bench.py (a prove starts at the first
poly::spmv_kernel of its pair; k = 1 is the last prove, k counts back like tools/gap_analysis.py):
python tools/prove_timeline.py <kernel_trace.csv> [k]   ->   start offset, duration, idle gap before it (us), queue, kernel"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "spmv" in r["Kernel_Name"]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 1
pr = rows[starts[-2 * k]:(starts[-2 * (k - 1)] if k > 1 else len(rows))]
t0 = int(pr[0]["Start_Timestamp"]); latest = t0; busy = 0; idle = 0
for r in pr:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    g = max(s - latest, 0); idle += g; busy += e - s
    print("%10.1f  dur %9.1f  gap %8.1f  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, g / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"].split("(")[0][-70:]))
    latest = max(latest, e)
print("span %.1f us, kernels %.1f us, idle %.1f us, %d dispatches" % ((latest - t0) / 1e3, busy / 1e3, idle / 1e3, len(pr)))
