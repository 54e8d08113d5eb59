cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03r; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_marlin.py tests/test_gpu_msm.py -m gpu -x -q -p no:cacheprovider -k "rccl or alternative_paths or selftest or sharded_prove" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log ); tail -5 $O/pytest.log
B="timeout 300 python bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-seam-route"
$B --simulate-rank 5/8 > $O/sim_5_8.json 2>/dev/null
$B --simulate-rank 3/4 > $O/sim_3_4.json 2>/dev/null
MH_FB_QUAD=0 $B --simulate-rank 3/4 > $O/sim_3_4_quad_off.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03r/sim*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['ms_per_step'], d['breakdown_ms_per_step'])
    except Exception as e: print(f,'ERR',e)
PY
