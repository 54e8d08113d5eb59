"""Host-side view of ONE prove from a rocprofv3 --hip-runtime-trace --kernel-trace run of bench.py: the HIP API calls of the calling
thread in order, with the time the host spent OUTSIDE the runtime before each call (= its own computation) and inside it.
python tools/host_gaps.py <hip_api_trace.csv> <kernel_trace.csv> [k] [min_us]   (k counts proves back from the end like gap_analysis.py)"""
import csv, sys
api = list(csv.DictReader(open(sys.argv[1])))
ker = list(csv.DictReader(open(sys.argv[2])))
k = int(sys.argv[3]) if len(sys.argv) > 3 else 1
min_us = float(sys.argv[4]) if len(sys.argv) > 4 else 15.0
ker.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(ker) if "spmv" in r["Kernel_Name"]]
t_begin = int(ker[starts[-2 * k]]["Start_Timestamp"])
t_end = int(ker[starts[-2 * (k - 1)]]["Start_Timestamp"]) if k > 1 else int(ker[-1]["End_Timestamp"])
api.sort(key=lambda r: int(r["Start_Timestamp"]))
tid_count = {}
for r in api:
    if t_begin <= int(r["Start_Timestamp"]) <= t_end:
        tid_count[r["Thread_Id"]] = tid_count.get(r["Thread_Id"], 0) + 1
main = max(tid_count, key=tid_count.get)
rows = [r for r in api if r["Thread_Id"] == main and t_begin - 2_000_000 <= int(r["Start_Timestamp"]) <= t_end]
prev_end = None
tot_out = tot_in = 0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    out = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    inside = (e - s) / 1e3
    if s >= t_begin:
        tot_out += max(out, 0); tot_in += inside
    if out >= min_us or inside >= min_us:
        print("%10.1f  host %8.1f us before, %8.1f us inside  %s" % ((s - t_begin) / 1e3, out, inside, r["Function"]))
    prev_end = e
print("prove span %.1f us: host outside the runtime %.1f us, inside HIP calls %.1f us, %d calls" % ((t_end - t_begin) / 1e3, tot_out, tot_in, len(rows)))
