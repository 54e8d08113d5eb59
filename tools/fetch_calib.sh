#!/bin/bash
# On the GPU box, from the repo root: tools/fetch_calib.sh <outdir>
# rocprofv3 --pmc FETCH_SIZE over tools/fetch_calib (known byte counts) -> <outdir>/fetch_calib.json with the factor
# known_bytes / (FETCH_SIZE KiB * 1024) per access pattern.
set -u
REPO=$(pwd)
OUT=${1:-gpurun_out/fetch_calib}; case $OUT in /*) ;; *) OUT=$REPO/$OUT;; esac; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
$REPO/tools/fetch_calib 2048 16777216 > $OUT/plain.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc -o pmc -- $REPO/tools/fetch_calib 2048 16777216 > $OUT/pmc.log 2>&1
cd $REPO
find $OUT/pmc -name "*counter_collection.csv" -exec cp {} $OUT/pmc_fetch_calib.csv \;
python3 - "$OUT" <<'PY'
import csv, json, sys, collections
out = sys.argv[1]
known = {"gather128_kernel": 16777216 * 128, "gather128_list_kernel": 16777216 * 132, "stream16_kernel": 16777216 * 128}
vals = collections.defaultdict(list)
for r in csv.DictReader(open(out + "/pmc_fetch_calib.csv")):
    if r.get("Counter_Name") == "FETCH_SIZE":
        vals[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
res = {}
for k, kb in known.items():
    v = vals.get(k)
    if v:
        res[k] = {"launches": len(v), "FETCH_SIZE_KiB_per_launch": sum(v) / len(v), "known_bytes": kb,
                  "factor_known_over_counter": kb / (sum(v) / len(v) * 1024.0)}
res["note"] = ("factor = known bytes / (FETCH_SIZE x 1024): what the counter must be multiplied by for this access pattern on gfx950. "
               "gather128 = one 128-B point per lane from a random slot of a 2 GiB table (8 x dwordx4), the pattern of "
               "msmfb::accum30_kernel; stream16 = the coalesced 16 B/lane pattern MI355X_MICROARCH.md's x2 refers to")
json.dump(res, open(out + "/fetch_calib.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
rm -rf $OUT/pmc
