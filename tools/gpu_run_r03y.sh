cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03y; mkdir -p $O
B="timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-seam-route --no-verify"
run() { tag=$1; shift; "$@" > $O/$tag.json 2>/dev/null; python - "$O/$tag.json" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split('/')[-1], d['ms_per_step'], d['breakdown_ms_per_step'])
except Exception as e: print(sys.argv[1],'ERR',e)
PY
}
run g1_prof_all   $B
run g1_prof_acc   env MH_PROF_MASK=4 $B
run g1_prof_all_b $B
run g1_prof_acc_b env MH_PROF_MASK=4 $B
run s58_prof_all   $B --simulate-rank 5/8
run s58_prof_acc   env MH_PROF_MASK=4 $B --simulate-rank 5/8
run s58_prof_all_b $B --simulate-rank 5/8
run s58_prof_acc_b env MH_PROF_MASK=4 $B --simulate-rank 5/8
run s34_prof_all   $B --simulate-rank 3/4
run s34_prof_acc   env MH_PROF_MASK=4 $B --simulate-rank 3/4
