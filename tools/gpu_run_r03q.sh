cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03q; mkdir -p $O
export TMPDIR=/tmp
for q in 0 1 2; do
  echo "== MH_FB_QUAD=$q"
  MH_FB_QUAD=$q timeout 600 python -m pytest tests/test_gpu_msm.py -m gpu -q -p no:cacheprovider -k "fixed_base_every_layout or selftest" 2>&1 | tail -12
done
