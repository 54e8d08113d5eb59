#!/bin/bash
# Usage (on the GPU box, from the repo root): tools/sweep_env.sh <tag> <VAR> <value> [<value> ...] [-- bench args]
# One short bench line per value of an environment variable of the library, in the order given (repeat a value to see the
# box's drift).  Writes gpurun_out/<tag>/sweep_<VAR>.txt.
set -u
TAG=${1:?tag}; VAR=${2:?variable}; shift 2
VALS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do VALS+=("$1"); shift; done; [ "${1:-}" = "--" ] && shift
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/$TAG; mkdir -p $O
for V in "${VALS[@]}"; do
  env $VAR=$V timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-throughput --no-seam-route --no-verify "$@" 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); b = d['breakdown_ms_per_step']
print('$VAR=$V', d['ms_per_step'], 'ms; accumulate', b['msm_accum'], 'sort+reduce', b['msm_sort_and_reduce_stages'], 'ntt', b['ntt'], '+', b.get('ntt_beside_msm_reduction'), 'beside the reduction',
      'golden', (d['proof'].get('oracle_golden') or {}).get('byte_identical'))"
done | tee $O/sweep_$VAR.txt
