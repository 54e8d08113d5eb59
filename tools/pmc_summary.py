"""Aggregate rocprofv3 counter_collection CSVs per kernel: FETCH_SIZE / WRITE_SIZE (KiB units per
the guide; FETCH_SIZE doubled for gfx950's wide-read under-count, MI355X_MICROARCH.md §HBM).

The x2 of the guide is calibrated on coalesced 16 B/lane streams; for the accumulate kernel's access pattern -- one random
128-byte table point per lane -- the factor measured by tools/fetch_calib.sh (profiles/*fetch_calib.json,
gather128_list_kernel) is used instead when that file is given as argv[2] (or found as profiles/fetch_calib.json).
argv[3] / LAUNCHES_PER_PROVE: accumulate launches of one prove (4 groups: rounds 1-3 and the openings)."""
import csv, json, sys, os, collections

d = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
calib_path = sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] else os.path.join(ROOT, "profiles", "fetch_calib.json")
GATHER_FACTOR, GATHER_SRC = 2.0, "MI355X_MICROARCH.md stream factor x2 (uncalibrated for gathers)"
if os.path.exists(calib_path):
    try:
        cj = json.load(open(calib_path))
        GATHER_FACTOR = float(cj["gather128_list_kernel"]["factor_known_over_counter"])
        GATHER_SRC = "%s: known bytes / FETCH_SIZE of a 128-B random gather per lane" % os.path.relpath(calib_path, ROOT)
    except Exception:
        pass
LAUNCHES = int(sys.argv[3]) if len(sys.argv) > 3 else int(os.environ.get("LAUNCHES_PER_PROVE", "4"))
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


for name in ("accum30", "msm::accum_kernel"):
    f = last_n("pmc_fetch.csv", "FETCH_SIZE", name, LAUNCHES)
    w = last_n("pmc_write.csv", "WRITE_SIZE", name, LAUNCHES)
    if f is not None and w is not None:
        out["prove_only:" + name] = {"dispatches": LAUNCHES, "fetch_bytes_per_launch_x2corrected": f * 1024 * 2,
                                     "fetch_bytes_per_launch_raw": f * 1024,
                                     "fetch_bytes_per_launch_gather_calibrated": f * 1024 * GATHER_FACTOR,
                                     "fetch_size_factor": GATHER_FACTOR, "fetch_size_factor_source": GATHER_SRC,
                                     "write_bytes_per_launch": w * 1024,
                                     "hbm_bytes_per_launch": f * 1024 * GATHER_FACTOR + w * 1024}
print(json.dumps(out, indent=1, sort_keys=True))
