#!/bin/bash
# Usage (GPU box): tools/ab.sh <other .so> [bench args]: alternate the in-tree build and another build of the same ABI on ONE box
set -u
OTHER=$1; shift
for i in $(seq 1 ${AB_ROUNDS:-3}); do
  for v in new old; do
    if [ $v = old ]; then export MARLIN_AMD_LIB=$OTHER; else unset MARLIN_AMD_LIB; fi
    timeout 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-throughput "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'], d['breakdown_ms_per_step'])"
  done
done
