#!/bin/bash
# Usage (on the GPU box, from the repo root): tools/rehearse_ranks.sh <tag> [log_constraints] [worlds...]
# The driver's multi-GPU launch rehearsed on ONE GPU at full size: `bench.py --gpus N` with all N ranks on device 0
# (BENCH_SINGLE_DEVICE=1, gloo for the set-up collectives), once through the native transport with the shared-memory stand-in for
# librccl (RCCL refuses two ranks on one device) and once through the torch.distributed callbacks.  The times mean nothing (N
# ranks share one GPU); REHEARSE_ARGS adds bench arguments (e.g. "--pc sonic").  What is checked is that every rank ends with the one-GPU proof at the headline size.
set -u
TAG=${1:?tag}; LOGN=${2:-20}; shift; shift || true
WORLDS=${*:-"8 4 2"}
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/$TAG; mkdir -p $O
for T in ${REHEARSE_TRANSPORTS:-native callback}; do
  for N in $WORLDS; do
    F=$O/bench_${N}_ranks_on_one_gpu_2p${LOGN}_${T}.json
    BENCH_BACKEND=gloo BENCH_SINGLE_DEVICE=1 MH_MOCK_RCCL_SLOT_MB=${MH_MOCK_RCCL_SLOT_MB:-1024} MH_RCCL_LIB=$PWD/tests/mock_rccl/libmock_rccl.so timeout 380 python bench.py --gpus $N --rehearsal --steps 3 --warmup 1 \
      --transport $T --log-constraints $LOGN --no-cpu-baseline --no-throughput ${REHEARSE_ARGS:-} > $F 2> $O/err_${N}_${T}.txt
    python - $F $N $T <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); p = d["proof"]
    print("ranks", sys.argv[2], sys.argv[3], "->", d["transport"]["kind"], "| sha256_32", p["sha256_32"], "| identical_on_all_ranks", p["identical_on_all_ranks"],
          "| golden", (p.get("oracle_golden") or {}).get("byte_identical"), "| verified", p["verified"], "|", d["config"]["parallelism"][:60])
except Exception as e:
    print("ranks", sys.argv[2], sys.argv[3], "FAILED", e)
PY
  done
done | tee $O/summary.txt
