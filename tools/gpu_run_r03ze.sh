cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03ze; mkdir -p $O
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests/test_gpu_msm.py tests/test_gpu_marlin.py tests/test_gpu_parity_pins.py -m gpu -x -q -p no:cacheprovider -k "not whole_proof_pinned" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log ); tail -4 $O/pytest.log
B="timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-seam-route"
run() { tag=$1; shift; "$@" > $O/$tag.json 2>/dev/null; python - "$O/$tag.json" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split('/')[-1], d['ms_per_step'], {k:v for k,v in d['breakdown_ms_per_step'].items() if k!='measured_on'}, (d.get('proof') or {}).get('verified'))
except Exception as e: print(sys.argv[1],'ERR',e)
PY
}
run g1_new   $B
run g1_old    env MARLIN_AMD_LIB=$GRAFT_REPO_ROOT/gpurun_prev/libmarlin_hip.so $B
run g1_new_b $B
run g1_old_b  env MARLIN_AMD_LIB=$GRAFT_REPO_ROOT/gpurun_prev/libmarlin_hip.so $B
run s58_new  $B --simulate-rank 5/8
run s58_old   env MARLIN_AMD_LIB=$GRAFT_REPO_ROOT/gpurun_prev/libmarlin_hip.so $B --simulate-rank 5/8
run s58_new_b  $B --simulate-rank 5/8
run s58_old_b   env MARLIN_AMD_LIB=$GRAFT_REPO_ROOT/gpurun_prev/libmarlin_hip.so $B --simulate-rank 5/8
