cd ${GRAFT_REPO_ROOT:-.}
B="timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-throughput --no-seam-route"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=d["breakdown_ms_per_step"]; print(d["ms_per_step"], "accum", b["msm_accum"], "host", b["host_and_other"], "glue", b["glue"])'
for rep in 1 2 3; do
for cfg in "--log-constraints 16 --pc sonic" "--pc sonic"; do
  for v in new old; do echo -n "$v $cfg: "; ( [ $v = old ] && export MARLIN_AMD_LIB=marlin_amd/libmarlin_hip_r06za.so; $B $cfg 2>/dev/null | python -c "$P" ); done
done; done
