set -x
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03i; mkdir -p $O
export TMPDIR=/tmp
B="timeout 300 python bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-seam-route"
for rg in 1/2 3/4 5/8; do n=$(echo $rg | tr '/' '_');
  $B --simulate-rank $rg > $O/sim_sliced_$n.json 2>/dev/null
  MH_SLICED_OPEN=0 $B --simulate-rank $rg > $O/sim_sliced_noopen_$n.json 2>/dev/null
  $B --simulate-rank $rg --no-sliced > $O/sim_replicated_$n.json 2>/dev/null
done
$B --log-constraints 22 --simulate-rank 3/8 > $O/sim_sliced_3_8_2p22.json 2>/dev/null
MH_SLICED_OPEN=0 $B --log-constraints 22 --simulate-rank 3/8 > $O/sim_sliced_noopen_3_8_2p22.json 2>/dev/null
$B --log-constraints 22 --simulate-rank 3/8 --no-sliced > $O/sim_replicated_3_8_2p22.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03i/sim_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['ms_per_step'], d['breakdown_ms_per_step'])
    except Exception as e: print(f,'ERR',e)
PY
