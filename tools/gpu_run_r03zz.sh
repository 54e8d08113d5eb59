set -x
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03zz; mkdir -p $O
export TMPDIR=/tmp
( timeout 2400 python -m pytest tests -m gpu -x -q --durations=4 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log ); tail -8 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke(); print(\"smoke ok\")" 2>&1 | tail -1
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
timeout 1200 bash tools/profile.sh r03zz > $O/profile.log 2>&1; tail -2 $O/profile.log
B="timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-seam-route"
$B --simulate-rank 1/2 > $O/sim_1_2.json 2>/dev/null
$B --simulate-rank 3/4 > $O/sim_3_4.json 2>/dev/null
$B --simulate-rank 5/8 > $O/sim_5_8.json 2>/dev/null
$B --log-constraints 22 --simulate-rank 3/8 > $O/sim_3_8_2p22.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03zz/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['ms_per_step'], d['value'], {k:v for k,v in d['breakdown_ms_per_step'].items() if k!='measured_on'}, d['roofline']['avg_launch_ms'], (d.get('proof') or {}).get('verified'), ((d.get('proof') or {}).get('oracle_golden') or {}).get('byte_identical'))
    except Exception as e: print(f,'ERR',e)
PY
