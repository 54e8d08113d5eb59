"""Idle time between consecutive dispatches of the LAST prove in a rocprofv3 --kernel-trace CSV of bench.py:
python tools/gap_analysis.py <kernel_trace.csv> [k].  A prove starts at the first poly::spmv_kernel of its pair; k = 1 (default)
analyses the last prove, k = 3 the third from the end (bench.py appends `breakdown_steps` proofs with events around every kernel
family after the timed ones: with --steps 2 the timed proofs are k = 3 and 4)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "spmv" in r["Kernel_Name"]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 1
pr = rows[starts[-2 * k]:(starts[-2 * (k - 1)] if k > 1 else len(rows))]
# Dispatches overlap since round 4 (copies on the copy stream beside the split kernel, transforms on the second stream beside the
# bucket reduction): a gap is time in which NOTHING runs -- from the latest end seen so far to the next start.
gaps = []
latest_end, latest = int(pr[0]["End_Timestamp"]), pr[0]
for b in pr[1:]:
    g = (int(b["Start_Timestamp"]) - latest_end) / 1e3      # us
    gaps.append((max(g, 0.0), latest["Kernel_Name"].split("(")[0][-40:], b["Kernel_Name"].split("(")[0][-40:]))
    if int(b["End_Timestamp"]) > latest_end:
        latest_end, latest = int(b["End_Timestamp"]), b
tot = sum(g for g, _, _ in gaps)
span = (int(pr[-1]["End_Timestamp"]) - int(pr[0]["Start_Timestamp"])) / 1e3
print("dispatches %d, span %.1f us, idle between dispatches %.1f us" % (len(pr), span, tot))
for lo, hi in ((0, 5), (5, 15), (15, 40), (40, 100), (100, 1e9)):
    sel = [g for g, _, _ in gaps if lo <= g < hi]
    print("  gaps in [%g, %g) us: %4d, sum %.1f us" % (lo, hi, len(sel), sum(sel)))
print("largest gaps (us, after kernel -> before kernel):")
for g, a, b in sorted(gaps, reverse=True)[:25]:
    print("  %8.1f  %s -> %s" % (g, a, b))

# the same gaps in dispatch order with their neighbourhood (which phase of the prove each host round trip belongs to)
print("gaps >= 15 us in order (index, us, two kernels before -> two after):")
short = lambda r: r["Kernel_Name"].split("(")[0].replace("void ", "")[-34:]
latest_end = int(pr[0]["End_Timestamp"])
for i, (a, b) in enumerate(zip(pr, pr[1:])):
    latest_end = max(latest_end, int(a["End_Timestamp"]))
    g = (int(b["Start_Timestamp"]) - latest_end) / 1e3
    if g >= 15:
        before = " | ".join(short(r) for r in pr[max(0, i - 1):i + 1])
        after = " | ".join(short(r) for r in pr[i + 1:i + 3])
        t = (int(a["End_Timestamp"]) - int(pr[0]["Start_Timestamp"])) / 1e3
        print("  #%3d  t=%8.1f us  gap %6.1f   %s  ->  %s" % (i, t, g, before, after))
