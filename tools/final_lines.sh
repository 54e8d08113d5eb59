#!/bin/bash
# Usage (GPU box, repo root): tools/final_lines.sh <tag> -- bench lines of the other configurations and of simulated ranks
set -u
O=gpurun_out/$1; mkdir -p $O
B="timeout 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-seam-route"
$B --simulate-rank 1/2 > $O/sim_rank_1_2.json 2>/dev/null
$B --simulate-rank 3/4 > $O/sim_rank_3_4.json 2>/dev/null
$B --simulate-rank 5/8 > $O/sim_rank_5_8.json 2>/dev/null
$B --log-constraints 18 > $O/bench_marlin_prove_2p18.json 2>/dev/null
$B --log-constraints 22 > $O/bench_marlin_prove_2p22.json 2>/dev/null
$B --log-constraints 22 --simulate-rank 3/8 > $O/sim_rank_3_8_2p22.json 2>/dev/null
$B --pc sonic > $O/bench_bls12_381_sonickzg10_2p20.json 2>/dev/null
MARLIN_AMD_CURVE=bn254 $B > $O/bench_bn254_marlinkzg10_2p20.json 2>/dev/null
MARLIN_AMD_CURVE=bn254 $B --pc sonic > $O/bench_bn254_sonickzg10_2p20.json 2>/dev/null
for f in $O/*.json; do python -c "
import sys,json
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], d['ms_per_step'], d['value'])
except Exception as e: print('$f','ERR',e)
"; done
