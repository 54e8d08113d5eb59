set -x
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03m; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; tail -c 3000 $O/bench_default.json
timeout 1200 bash tools/profile.sh r03m > $O/profile.log 2>&1; tail -3 $O/profile.log
cat gpurun_out/prof_r03m/pmc_traffic.json | head -30
( timeout 600 python -m pytest tests/test_gpu_marlin.py -m gpu -x -q -p no:cacheprovider -k "sharded or bench_gpus" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log ); tail -4 $O/pytest.log
