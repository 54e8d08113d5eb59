#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export MASTER_ADDR=127.0.0.1 WORLD_SIZE=8
for rep in $(seq 1 ${2:-12}); do
  export MASTER_PORT=$((29877 + rep))
  for r in 0 1 2 3 4 5 6 7; do RANK=$r timeout 300 python tools/scratch/r05_stress8.py ${1:-2} 2>&1 | grep -E "^rank.*(DIFF|bad [1-9])|Error" & done; wait
  echo "rep $rep done"
done
