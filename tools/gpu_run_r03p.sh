set -x
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03p; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_msm.py -m gpu -x -q -p no:cacheprovider > $O/pytest_msm.log 2>&1; echo "pytest rc=$?" >> $O/pytest_msm.log ); tail -4 $O/pytest_msm.log
( timeout 900 python -m pytest tests/test_gpu_marlin.py tests/test_gpu_parity_pins.py -m gpu -x -q -p no:cacheprovider -k "sharded or golden or skew or 2p18 or 18" > $O/pytest_marlin.log 2>&1; echo "pytest rc=$?" >> $O/pytest_marlin.log ); tail -4 $O/pytest_marlin.log
B="timeout 300 python bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-seam-route"
run() { tag=$1; shift; "$@" > $O/$tag.json 2>/dev/null; python - "$O/$tag.json" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split('/')[-1], d['ms_per_step'], d['breakdown_ms_per_step'])
except Exception as e: print(sys.argv[1],'ERR',e)
PY
}
run g1_quad_off   env MH_FB_QUAD=0 $B
run g1_quad_auto  $B
run g1_quad_both  env MH_FB_QUAD=2 $B
run g1_quad_both_131k env MH_FB_QUAD=2 MH_FB_SEG_THREADS=131072 $B
run g1_quad_off_b env MH_FB_QUAD=0 $B
run g1_quad_auto_b $B
run s58_quad_off  env MH_FB_QUAD=0 $B --simulate-rank 5/8
run s58_quad_r2   env MH_FB_QUAD=1 $B --simulate-rank 5/8
run s58_quad_auto $B --simulate-rank 5/8
run s58_quad_32k  env MH_FB_SEG_THREADS=32768 $B --simulate-rank 5/8
run s58_quad_131k env MH_FB_SEG_THREADS=131072 $B --simulate-rank 5/8
run s34_quad_off  env MH_FB_QUAD=0 $B --simulate-rank 3/4
run s34_quad_auto $B --simulate-rank 3/4
run s34_quad_131k env MH_FB_SEG_THREADS=131072 $B --simulate-rank 3/4
run s12_quad_auto $B --simulate-rank 1/2
run s12_quad_both env MH_FB_QUAD=2 $B --simulate-rank 1/2
run s38_22_quad_off env MH_FB_QUAD=0 $B --log-constraints 22 --simulate-rank 3/8
run s38_22_quad_auto $B --log-constraints 22 --simulate-rank 3/8
