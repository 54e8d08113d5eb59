set -x
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03o; mkdir -p $O
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_parity_pins.py -m gpu -x -q -p no:cacheprovider -k "golden_large" > $O/pytest_golden_xl.log 2>&1; echo "pytest rc=$?" >> $O/pytest_golden_xl.log ); tail -4 $O/pytest_golden_xl.log
# which kernels make up one rank's share at 8 ranks: kernel trace of a simulated rank
cd /tmp
for cfg in "5_8:--simulate-rank 5/8" "3_8_2p22:--log-constraints 22 --simulate-rank 3/8"; do
  tag=${cfg%%:*}; args=${cfg#*:}
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$tag -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-seam-route $args > $O/trace_$tag.log 2>&1
  find $O/trace_$tag -name "*kernel_stats.csv" -exec cp {} $O/sim_${tag}_kernel_stats.csv \;
  python3 - "$O" "$tag" <<'PY'
import csv, glob, sys, collections
out, tag = sys.argv[1], sys.argv[2]
rows = []
for f in glob.glob(out + "/trace_" + tag + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# the last prove = the dispatches after the last-but-one occurrence pattern: take the last third of accum dispatches' window
acc = [i for i, r in enumerate(rows) if "accum30_kernel" in r[2]]
per = 4
first = acc[-per]            # first accumulate launch of the last prove
# walk back to the previous accumulate's end + its reduce: the last prove starts after the previous prove's last kernel; use time gap heuristic
start_i = acc[-per - 1] + 1 if len(acc) > per else 0
agg = collections.OrderedDict()
for s, e, k in rows[start_i:]:
    k = k.split("(")[0][:70]
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += (e - s) / 1e6
tot = sum(v[1] for v in agg.values())
span = (rows[-1][1] - rows[start_i][0]) / 1e6
with open(out + "/sim_" + tag + "_last_prove_kernels.txt", "w") as f:
    f.write("kernels from the dispatch after the previous prove's last accumulate to the end (includes that prove's reduce tail): busy %.3f ms, span %.3f ms\n" % (tot, span))
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        f.write("%9.3f ms %5d  %s\n" % (v[1], v[0], k))
PY
  head -30 $O/sim_${tag}_last_prove_kernels.txt
  rm -rf $O/trace_$tag
done
cd $GRAFT_REPO_ROOT
MH_TRACE=1 timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-seam-route --simulate-rank 5/8 > $O/trace_phases_5_8.json 2> $O/trace_phases_5_8.txt
tail -60 $O/trace_phases_5_8.txt
