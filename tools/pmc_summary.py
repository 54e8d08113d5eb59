"""Aggregate rocprofv3 counter_collection CSVs per kernel: FETCH_SIZE / WRITE_SIZE (KiB units per
the guide; FETCH_SIZE doubled for gfx950's wide-read under-count, MI355X_MICROARCH.md §HBM)."""
import csv, json, sys, os, collections

d = sys.argv[1]
res = collections.defaultdict(lambda: {"launches": 0, "FETCH_SIZE_KiB": 0.0, "WRITE_SIZE_KiB": 0.0})
for fn, key in (("pmc_fetch.csv", "FETCH_SIZE"), ("pmc_write.csv", "WRITE_SIZE")):
    p = os.path.join(d, fn)
    if not os.path.exists(p):
        continue
    seen = collections.Counter()
    with open(p) as f:
        for row in csv.DictReader(f):
            if row.get("Counter_Name") != key:
                continue
            k = row["Kernel_Name"].split("(")[0]
            res[k][key + "_KiB"] += float(row["Counter_Value"])
            seen[k] += 1
    for k, n in seen.items():
        res[k]["launches"] = max(res[k]["launches"], n)
out = {}
for k, v in res.items():
    n = max(1, v["launches"])
    fetch_b = v["FETCH_SIZE_KiB"] * 1024 * 2      # gfx950: FETCH_SIZE reports 1/2 of wide coalesced reads
    write_b = v["WRITE_SIZE_KiB"] * 1024
    out[k] = {"launches": v["launches"], "fetch_bytes_per_launch_x2corrected": fetch_b / n,
              "fetch_bytes_per_launch_raw": v["FETCH_SIZE_KiB"] * 1024 / n,
              "write_bytes_per_launch": write_b / n, "hbm_bytes_per_launch": (fetch_b + write_b) / n}
# prove-only view of the dominant kernel: `bench.py --steps 1 --warmup 0` launches the accumulate kernel 6 times inside
# Marlin::index (larger batches) and then 4 times inside the one timed prove; the LAST 4 dispatches are the prove's
def last_n(fn, key, name, n):
    p = os.path.join(d, fn)
    rows = []
    if os.path.exists(p):
        with open(p) as f:
            for row in csv.DictReader(f):
                if row.get("Counter_Name") == key and name in row["Kernel_Name"]:
                    rows.append((int(row.get("Dispatch_Id", len(rows))), float(row["Counter_Value"])))
    rows.sort()
    vals = [v for _, v in rows[-n:]]
    return sum(vals) / len(vals) if vals else None


for name in ("accum30_kernel", "msm::accum_kernel"):
    f = last_n("pmc_fetch.csv", "FETCH_SIZE", name, 4)
    w = last_n("pmc_write.csv", "WRITE_SIZE", name, 4)
    if f is not None and w is not None:
        out["prove_only:" + name] = {"dispatches": 4, "fetch_bytes_per_launch_x2corrected": f * 1024 * 2,
                                     "write_bytes_per_launch": w * 1024, "hbm_bytes_per_launch": f * 1024 * 2 + w * 1024}
print(json.dumps(out, indent=1, sort_keys=True))
