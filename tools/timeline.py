"""Timeline of the kernels of the LAST prove in a rocprofv3 kernel trace: python tools/timeline.py <kernel_trace.csv> [n_last]
Prints per dispatch: start offset, duration, idle gap before it (ms); and the totals."""
import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]))
rows.sort()
# the last prove starts at the last chacha / first kernel after the longest idle gap near the end: take dispatches after the
# last gap > 2 ms (bench barrier) -- or the last n given
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else None
if n_last is None:
    cut = 0
    for i in range(1, len(rows)):
        if rows[i][0] - rows[i - 1][1] > 1_500_000:
            cut = i
    rows = rows[cut:]
else:
    rows = rows[-n_last:]
t0 = rows[0][0]
busy = 0
gaps = 0
prev_end = t0
for s, e, name in rows:
    gap = s - prev_end
    if gap > 0: gaps += gap
    busy += e - s
    print("%9.3f  dur %8.3f  gap %7.3f  %s" % ((s - t0) / 1e6, (e - s) / 1e6, gap / 1e6, name[-60:]))
    prev_end = max(prev_end, e)
print("span %.3f ms, kernels %.3f ms, idle gaps %.3f ms, %d dispatches" % ((prev_end - t0) / 1e6, busy / 1e6, gaps / 1e6, len(rows)))
