cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03z; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_marlin.py -m gpu -x -q -p no:cacheprovider -k "bench or rccl" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log ); tail -3 $O/pytest.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
timeout 900 python bench.py --steps 20 --warmup 5 --full-prof --no-cpu-baseline --no-seam-route > $O/bench_full_prof.json 2>/dev/null
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-seam-route > $O/bench_default_b.json 2>/dev/null
B="timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-seam-route"
$B --simulate-rank 5/8 > $O/sim_5_8.json 2>/dev/null
$B --simulate-rank 3/4 > $O/sim_3_4.json 2>/dev/null
$B --simulate-rank 1/2 > $O/sim_1_2.json 2>/dev/null
$B --log-constraints 22 --simulate-rank 3/8 > $O/sim_3_8_2p22.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03z/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['ms_per_step'], d['value'], {k:v for k,v in d['breakdown_ms_per_step'].items() if k!='measured_on'}, d['roofline']['avg_launch_ms'], (d.get('proof') or {}).get('verified'))
    except Exception as e: print(f,'ERR',e)
PY
