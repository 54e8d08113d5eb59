#!/bin/bash
# Usage (GPU box, repo root): tools/sim_profile.sh <tag> R/G [bench args]: kernel stats of one simulated rank (bench.py --simulate-rank)
set -u
R=$(pwd); TAG=$1; RG=$2; shift 2
O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
N=$(echo $RG | tr '/' '_')
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr_$N -o trace -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-throughput --simulate-rank $RG "$@" > $O/sim_$N.log 2>&1
find $O/tr_$N -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_sim_$N.csv \;
find $O/tr_$N -name "*kernel_trace.csv" -exec cp {} $O/kernel_trace_sim_$N.csv \;
rm -rf $O/tr_$N
tail -1 $O/sim_$N.log | cut -c1-300
