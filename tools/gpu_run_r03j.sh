set -x
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03j; mkdir -p $O
export TMPDIR=/tmp
B="timeout 300 python bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-seam-route"
for w in auto 2 3 4; do
  if [ $w = auto ]; then unset MH_ACC_WAVES; else export MH_ACC_WAVES=$w; fi
  $B --simulate-rank 5/8 > $O/sim_5_8_waves_$w.json 2>/dev/null
  $B --simulate-rank 3/4 > $O/sim_3_4_waves_$w.json 2>/dev/null
  $B --log-constraints 22 --simulate-rank 3/8 > $O/sim_3_8_2p22_waves_$w.json 2>/dev/null
done
unset MH_ACC_WAVES
$B > $O/bench_single.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03j/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['ms_per_step'], d['breakdown_ms_per_step'])
    except Exception as e: print(f,'ERR',e)
PY
