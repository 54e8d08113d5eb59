set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03a
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -x -q --durations=25 -p no:cacheprovider > gpurun_out/r03a/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03a/pytest.log ) 
tail -40 gpurun_out/r03a/pytest.log
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r03a/bench_split.json 2> gpurun_out/r03a/bench_split.err
MH_FB_SPLIT=0 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r03a/bench_nosplit.json 2> gpurun_out/r03a/bench_nosplit.err
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r03a/bench_split2.json 2>> gpurun_out/r03a/bench_split.err
MH_FB_SPLIT=0 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r03a/bench_nosplit2.json 2>> gpurun_out/r03a/bench_nosplit.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03a/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['ms_per_step'], d['breakdown_ms_per_step'])
    except Exception as e: print(f, 'ERR', e)
PY
