#!/bin/bash
# Usage (on the GPU box, from the repo root): tools/profile_sq.sh <tag> [bench args...]
# SQ / GRBM counters of one kernel over ONE prove (default: the accumulate kernel; SQ_KERNEL=rsum_kernel: the
# row / column sums of the bucket reduction): rocprofv3 --pmc in its own run (with
# --kernel-trace only), per dispatch; the summary keeps the prove's own dispatches (the last 4 launches of
# msmfb::accum30v_kernel: commit rounds 1-3 and the openings) apart from Marlin::index's.
#   gpurun_out/prof_<tag>/sq_counters.json
set -u
TAG=${1:-r02}; shift || true
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
CMD1="python $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-throughput $*"
# SQ_COUNTERS overrides the 8 SQ slots; the default set is the wave-cycle breakdown of MI355X_MICROARCH.md (rocprofv3 PMC slots):
# WAIT_ANY (wave parked: s_waitcnt / barrier) + WAIT_INST_ANY (issue stall) + ACTIVE_INST_ANY ~ WAVE_CYCLES, all in quad-cycles
SQ_COUNTERS=${SQ_COUNTERS:-"SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES"}
timeout 900 rocprofv3 --pmc $SQ_COUNTERS GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d $OUT/pmc_sq -o pmc -- $CMD1 > $OUT/pmc_sq.log 2>&1
find $OUT/pmc_sq -name "*counter_collection.csv" -exec cp {} $OUT/pmc_sq.csv \;
cd $REPO
python3 - "$OUT" "${SQ_KERNEL:-accum30v_kernel}" <<'PY'
import csv, json, sys, collections
out, kname = sys.argv[1], sys.argv[2]
rows = collections.defaultdict(dict)          # dispatch id -> counters
names = {}
dur = {}
for r in csv.DictReader(open(out + "/pmc_sq.csv")):
    d = int(r["Dispatch_Id"])
    rows[d][r["Counter_Name"]] = rows[d].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    names[d] = r["Kernel_Name"]
    dur[d] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
acc = [d for d in sorted(rows) if kname in names[d]]
def summarise(ds):
    tot = collections.Counter()
    for d in ds:
        for k, v in rows[d].items():
            tot[k] += v
    ms = sum(dur.get(d, 0.0) for d in ds)
    s = {"dispatches": len(ds), "kernel_ms_under_pmc": round(ms, 3)}
    s.update({k: v for k, v in tot.items()})
    if tot.get("GRBM_GUI_ACTIVE") and ms:
        s["effective_clock_GHz"] = round(tot["GRBM_GUI_ACTIVE"] / (ms * 1e6), 3)     # the counter is summed over dispatches, 1 per cycle
    if tot.get("SQ_INSTS_VALU") and ms:
        s["valu_wave_instr_per_s_T"] = round(tot["SQ_INSTS_VALU"] / (ms * 1e-3) / 1e12, 4)
        s["valu_lane_instr_per_s_T"] = round(64 * tot["SQ_INSTS_VALU"] / (ms * 1e-3) / 1e12, 3)
    if tot.get("SQ_ACTIVE_INST_VALU") and tot.get("SQ_INSTS_VALU"):
        # SQ_ACTIVE_INST_VALU counts quad-cycles (MI355X_MICROARCH.md): x4 = cycles a SIMD spent issuing VALU
        s["valu_issue_cycles_per_wave_instr"] = round(4 * tot["SQ_ACTIVE_INST_VALU"] / tot["SQ_INSTS_VALU"], 3)
    if tot.get("SQ_ACTIVE_INST_VALU") and tot.get("SQ_BUSY_CYCLES"):
        s["valu_active_over_busy"] = round(tot["SQ_ACTIVE_INST_VALU"] / tot["SQ_BUSY_CYCLES"], 4)
    if tot.get("SQ_WAVE_CYCLES"):
        wc = tot["SQ_WAVE_CYCLES"]
        s["wave_cycle_breakdown"] = {k: round(tot[c] / wc, 4) for k, c in (("parked_waitcnt_or_barrier", "SQ_WAIT_ANY"),
                                     ("issue_stalled", "SQ_WAIT_INST_ANY"), ("issuing_any", "SQ_ACTIVE_INST_ANY"),
                                     ("issuing_valu", "SQ_ACTIVE_INST_VALU")) if c in tot}
        s["wave_cycle_breakdown"]["unaccounted"] = round(1.0 - sum(tot.get(c, 0) for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY")) / wc, 4)
    if tot.get("SQ_WAVE_CYCLES") and tot.get("SQ_BUSY_CYCLES"):
        s["resident_waves_per_busy_simd_cycle"] = round(tot["SQ_WAVE_CYCLES"] / tot["SQ_BUSY_CYCLES"], 3)
    return s
res = {"kernel": "msmfb::" + kname, "all_dispatches": summarise(acc), "prove_only_last4": summarise(acc[-4:]),
       "per_dispatch": [dict(rows[d], dispatch=d, ms=dur.get(d)) for d in acc],
       "note": "counters summed over SEs/XCDs as rocprofv3 reports them; prove_only = the last 4 dispatches of the run "
               "(bench.py --steps 1 --warmup 0: index first, then one prove)"}
json.dump(res, open(out + ("/sq_counters.json" if kname.startswith("accum30") else "/sq_counters_%s.json" % kname), "w"), indent=1)
print(json.dumps({k: res[k] for k in ("all_dispatches", "prove_only_last4")}, indent=1))
PY
rm -rf $OUT/pmc_sq
ls -la $OUT
