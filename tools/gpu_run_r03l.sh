set -x
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03l; mkdir -p $O
export TMPDIR=/tmp
B="timeout 300 python bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-seam-route"
for w in auto 3 2; do
  if [ $w = auto ]; then unset MH_ACC_WAVES; else export MH_ACC_WAVES=$w; fi
  $B --simulate-rank 5/8 > $O/sim_5_8_waves_$w.json 2>/dev/null
  $B --log-constraints 22 --simulate-rank 3/8 > $O/sim_3_8_2p22_waves_$w.json 2>/dev/null
done
unset MH_ACC_WAVES
$B --simulate-rank 3/4 > $O/sim_3_4_waves_auto.json 2>/dev/null
( BENCH_BACKEND=gloo BENCH_SINGLE_DEVICE=1 timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 --log-constraints 18 --no-cpu-baseline > $O/bench_gpus2_gloo_2p18.json 2> $O/bench_gpus2.err ); tail -3 $O/bench_gpus2.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03l/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['ms_per_step'], d['breakdown_ms_per_step'], d.get('config',{}).get('parallelism','')[:60])
    except Exception as e: print(f,'ERR',e)
PY
