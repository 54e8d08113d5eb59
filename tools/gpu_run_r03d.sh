set -x
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03d; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_dist_blocks.py tests/test_gpu_msm.py tests/test_gpu_marlin.py -m gpu -x -q -p no:cacheprovider -k "dist or msm or sharded or golden" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log ); tail -30 $O/pytest.log
timeout 1200 bash tools/profile.sh r03d > $O/profile.log 2>&1; tail -5 $O/profile.log
cat gpurun_out/prof_r03d/pmc_traffic.json gpurun_out/prof_r03d/accum_dispatches.json
