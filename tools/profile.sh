#!/bin/bash
# Usage (on the GPU box, from the repo root): tools/profile.sh <tag> [bench args...]
# Produces gpurun_out/prof_<tag>/{kernel_stats.csv,pmc_fetch.csv,pmc_write.csv,pmc_summary.json}
set -u
TAG=${1:-r01}; shift || true
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
CMD="python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-throughput --no-seam-route --full-prof $*"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
# per-dispatch durations of the accumulate kernel: kernel_stats averages over index + warm-up + timed launches, the
# bench's roofline over the timed ones only (the last 4 x steps dispatches)
python3 - "$OUT" <<'PY'
import csv, glob, json, sys
out = sys.argv[1]
rows = []
for f in glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "accum30" in r["Kernel_Name"] or "msm::accum_kernel" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
rows.sort()
d = [x[1] / 1e6 for x in rows]
steps = 2
per_prove = 4            # the 4 MSM groups of a prove (rounds 1-3, openings)
timed = d[-per_prove * steps:]
json.dump({"kernel": "bucket accumulation (msmfb::accum30v_kernel)", "dispatch_ms_in_launch_order": [round(x, 3) for x in d],
           "avg_ms_all_dispatches": round(sum(d) / max(1, len(d)), 3),
           "avg_ms_timed_region": round(sum(timed) / max(1, len(timed)), 3),
           "note": "timed region = the last 4 x %d dispatches (4 MSM groups per prove: rounds 1-3 and the openings); "
                   "earlier dispatches belong to Marlin::index (larger batches) and the warm-up prove" % steps},
          open(out + "/accum_dispatches.json", "w"), indent=1)
PY
grep '^{' $OUT/trace.log | tail -1 > $OUT/bench_line_under_rocprof.json
if [ "${PROFILE_STATS_ONLY:-0}" = 1 ]; then rm -rf $OUT/trace; ls -la $OUT; exit 0; fi
CMD1="python $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-throughput --no-seam-route --full-prof $*"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o pmc -- $CMD1 > $OUT/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o pmc -- $CMD1 > $OUT/pmc_write.log 2>&1
find $OUT/pmc_fetch -name "*counter_collection.csv" -exec cp {} $OUT/pmc_fetch.csv \;
find $OUT/pmc_write -name "*counter_collection.csv" -exec cp {} $OUT/pmc_write.csv \;
cd $REPO
LPP=$(python3 -c "import json,sys; print(int(json.loads([l for l in open('$OUT/pmc_fetch.log') if l.startswith('{')][-1])['accum_launches_per_step']))" 2>/dev/null || echo 8)
python tools/pmc_summary.py $OUT "" $LPP > $OUT/pmc_summary.json 2> $OUT/pmc_summary.err
python3 - "$OUT" "$TAG" "$LPP" <<'PY'
import hashlib, json, os, socket, subprocess, sys
out, tag, lpp = sys.argv[1], sys.argv[2], int(sys.argv[3])
sys.path.insert(0, os.getcwd())
from marlin_amd import workload as W
d = json.load(open(out + "/pmc_summary.json"))
po = d.get("prove_only:accum30")
if po:
    pairs = sum(n for n, _ in W.msm_executed(1 << 20))
    try:
        dev = subprocess.run(["rocm-smi", "--showproductname"], capture_output=True, text=True, timeout=20).stdout.strip().splitlines()
        dev = [l for l in dev if "Card Series" in l or "Card SKU" in l][:2]
    except Exception:
        dev = []
    json.dump({"source": "profiles/%s_pmc_summary_marlin_prove_2p20.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, of "
                         "`bench.py --steps 1 --warmup 0`; PROVE ONLY: the last %d dispatches of msmfb::accum30v_kernel = the launches "
                         "of the one timed prove)" % (tag, lpp),
               "box": {"hostname": socket.gethostname(), "device": dev},
               "build": {"libmarlin_hip.so_sha256_16": hashlib.sha256(open("marlin_amd/libmarlin_hip.so", "rb").read()).hexdigest()[:16]},
               "fetch_size_factor": po.get("fetch_size_factor"), "fetch_size_factor_source": po.get("fetch_size_factor_source"),
               "launches_per_prove": lpp, "pairs_per_prove": pairs,
               "msm_accum_bytes_per_launch": po["hbm_bytes_per_launch"],
               "msm_accum_bytes_per_pair": po["hbm_bytes_per_launch"] * lpp / pairs,
               "fetch_bytes_per_launch_raw": po.get("fetch_bytes_per_launch_raw"),
               "fetch_bytes_per_launch_x2corrected": po["fetch_bytes_per_launch_x2corrected"],
               "write_bytes_per_launch": po["write_bytes_per_launch"]}, open(out + "/pmc_traffic.json", "w"), indent=1)
PY
find $OUT -name "*.csv" | head -20; rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write
ls -la $OUT
