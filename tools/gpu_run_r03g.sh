set -x
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03g; mkdir -p $O
export TMPDIR=/tmp
for lg in 16 18; do for c in 13 14 15 16 17 18 19; do
  MH_FB_C=$c timeout 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-seam-route --log-constraints $lg 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('2^$lg c=$c', d['ms_per_step'], d['breakdown_ms_per_step'])" >> $O/window_sweep.txt 2>&1
done; done
for lg in 16; do for c in 13 14 15 16 17 18; do
  MH_FB_C=$c timeout 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-seam-route --log-constraints $lg --pc sonic 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sonic 2^$lg c=$c', d['ms_per_step'], d['breakdown_ms_per_step'])" >> $O/window_sweep.txt 2>&1
done; done
cat $O/window_sweep.txt
