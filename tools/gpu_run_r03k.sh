set -x
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03k; mkdir -p $O
export TMPDIR=/tmp
B="timeout 300 python bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-seam-route"
for rg in 1/2 3/4 5/8 0/8 7/8; do n=$(echo $rg | tr '/' '_'); $B --simulate-rank $rg > $O/sim_sliced_$n.json 2>/dev/null; done
$B --log-constraints 22 --simulate-rank 3/8 > $O/sim_sliced_3_8_2p22.json 2>/dev/null
$B --log-constraints 22 --simulate-rank 1/4 > $O/sim_sliced_1_4_2p22.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03k/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['ms_per_step'], d['breakdown_ms_per_step'])
    except Exception as e: print(f,'ERR',e)
PY
( timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log ); tail -14 $O/pytest.log
