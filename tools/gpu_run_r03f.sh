set -x
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03f; mkdir -p $O
export TMPDIR=/tmp
timeout 900 bash tools/ab.sh $GRAFT_REPO_ROOT/marlin_amd/csrc/build_alt/libmarlin_hip_alt.so --no-seam-route > $O/ab_reduce_inline.txt 2>&1; cat $O/ab_reduce_inline.txt
timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; python -c "
import json; d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['breakdown_ms_per_step']); print(d['seam_route'])"
( timeout 300 python -m pytest tests/test_gpu_msm.py -m gpu -x -q -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log ); tail -3 $O/pytest.log
