"""Per-kernel time of the LAST prove of a rocprofv3 --kernel-trace CSV of `bench.py` (a prove starts at the first
poly::spmv_kernel of its pair): python tools/prove_kernels.py <kernel_trace.csv> [filter]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "spmv" in r["Kernel_Name"]]
pr = rows[starts[-2]:]
t0, t1 = int(pr[0]["Start_Timestamp"]), int(pr[-1]["End_Timestamp"])
agg = collections.OrderedDict()
busy = 0
for r in pr:
    n = r["Kernel_Name"].split("(")[0][:48]
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    a = agg.setdefault(n, [0, 0]); a[0] += 1; a[1] += d; busy += d
print("span %.3f ms, busy %.3f ms, %d dispatches" % ((t1 - t0) / 1e6, busy / 1e6, len(pr)))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for n, (c, d) in sorted(agg.items(), key=lambda x: -x[1][1]):
    if flt in n:
        print("%-50s %4d %9.3f" % (n, c, d / 1e6))
