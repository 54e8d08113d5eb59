cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03za; mkdir -p $O
B="timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-seam-route"
run() { tag=$1; shift; "$@" > $O/$tag.json 2>/dev/null; python - "$O/$tag.json" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split('/')[-1], d['ms_per_step'], {k:v for k,v in d['breakdown_ms_per_step'].items() if k!='measured_on'})
except Exception as e: print(sys.argv[1],'ERR',e)
PY
}
run s58_sync    env BENCH_SIM_SYNC_EXCHANGE=1 $B --simulate-rank 5/8
run s58_ordered $B --simulate-rank 5/8
run s58_sync_b    env BENCH_SIM_SYNC_EXCHANGE=1 $B --simulate-rank 5/8
run s58_ordered_b $B --simulate-rank 5/8
run s34_sync    env BENCH_SIM_SYNC_EXCHANGE=1 $B --simulate-rank 3/4
run s34_ordered $B --simulate-rank 3/4
run s38_22_sync    env BENCH_SIM_SYNC_EXCHANGE=1 $B --log-constraints 22 --simulate-rank 3/8
run s38_22_ordered $B --log-constraints 22 --simulate-rank 3/8
