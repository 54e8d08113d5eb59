set -x
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03h; mkdir -p $O
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests/test_gpu_marlin.py tests/test_gpu_dist_blocks.py -m gpu -x -q -p no:cacheprovider -k "sharded_prove or dist" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log ); tail -30 $O/pytest.log
B="timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-seam-route"
for rg in 1/2 3/4 5/8; do n=$(echo $rg | tr '/' '_');
  $B --simulate-rank $rg > $O/sim_sliced_$n.json 2>/dev/null
  $B --simulate-rank $rg --no-sliced > $O/sim_replicated_$n.json 2>/dev/null
done
$B --log-constraints 22 --simulate-rank 3/8 > $O/sim_sliced_3_8_2p22.json 2>/dev/null
$B --log-constraints 22 --simulate-rank 3/8 --no-sliced > $O/sim_replicated_3_8_2p22.json 2>/dev/null
for f in $O/sim_*.json; do python -c "
import sys,json
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], d['ms_per_step'], d['breakdown_ms_per_step'])
except Exception as e: print('$f','ERR',e)
"; done
