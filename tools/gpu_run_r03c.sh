set -x
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03c; mkdir -p $O
export TMPDIR=/tmp
timeout 300 bash tools/fetch_calib.sh $O/fetch_calib > $O/fetch_calib.log 2>&1; tail -32 $O/fetch_calib.log
cp $O/fetch_calib/fetch_calib.json profiles/fetch_calib.json 2>/dev/null
timeout 1200 bash tools/profile.sh r03c > $O/profile.log 2>&1; tail -15 $O/profile.log
cat gpurun_out/prof_r03c/pmc_traffic.json
timeout 300 python bench.py --steps 5 --warmup 2 --workload seam-route > $O/bench_seam_route.json 2> $O/bench_seam_route.err; tail -c 600 $O/bench_seam_route.json
timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-seam-route --log-constraints 16 --pc sonic > $O/bench_reference_shape_2p16_sonic.json 2>/dev/null
( timeout 300 python -m pytest tests/test_gpu_marlin.py -m gpu -x -q -k "rccl or draws or skewed" -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log ); tail -5 $O/pytest.log
timeout 900 bash tools/final_lines.sh r03c_lines > $O/final_lines.log 2>&1; cat $O/final_lines.log
