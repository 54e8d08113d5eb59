#!/bin/bash
# Usage (on the GPU box, from the repo root): tools/profile.sh <tag> [bench args...]
# Produces gpurun_out/prof_<tag>/{kernel_stats.csv,pmc_fetch.csv,pmc_write.csv,pmc_summary.json}
set -u
TAG=${1:-r01}; shift || true
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
CMD="python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline $*"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
CMD1="python $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline $*"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o pmc -- $CMD1 > $OUT/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o pmc -- $CMD1 > $OUT/pmc_write.log 2>&1
find $OUT/pmc_fetch -name "*counter_collection.csv" -exec cp {} $OUT/pmc_fetch.csv \;
find $OUT/pmc_write -name "*counter_collection.csv" -exec cp {} $OUT/pmc_write.csv \;
cd $REPO
python tools/pmc_summary.py $OUT > $OUT/pmc_summary.json 2> $OUT/pmc_summary.err
find $OUT -name "*.csv" | head -20; rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write
ls -la $OUT
