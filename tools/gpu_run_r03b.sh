set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03b; mkdir -p $O
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_ntt.py tests/test_gpu_parity_pins.py -m gpu -x -q -k "not 22 and not 20" -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log )
tail -5 $O/pytest.log
timeout 300 python tools/ntt_bench.py 16 18 20 22 23 24 > $O/ntt_bench.txt 2>&1
cat $O/ntt_bench.txt
timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err
MH_ACC_WAVES=2 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-seam-route > $O/bench_waves2.json 2> $O/bench_waves2.err
MH_ACC_WAVES=2 MH_FB_SPLIT=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-seam-route > $O/bench_waves2_split.json 2> $O/bench_waves2_split.err
timeout 300 python bench.py --steps 5 --warmup 2 --workload seam-route > $O/bench_seam_route.json 2> $O/bench_seam_route.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03b/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['ms_per_step'], d['breakdown_ms_per_step'], d.get('seam_route'))
    except Exception as e: print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-500:])
PY
timeout 400 bash tools/fetch_calib.sh $O/fetch_calib > $O/fetch_calib.log 2>&1; tail -30 $O/fetch_calib.log
