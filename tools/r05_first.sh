#!/bin/bash
# round 5, first lease: the row/column + bit-plane bucket reduction -- correctness (MSM suite, marlin goldens), A/B against the round-4 build, kernel trace
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=$(pwd)/gpurun_out/${1:-r05a}; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_msm.py -x -q -p no:cacheprovider > $O/pytest_msm.log 2>&1; echo "rc=$?" >> $O/pytest_msm.log ); tail -4 $O/pytest_msm.log
( timeout 900 python -m pytest tests/test_gpu_marlin.py -x -q -p no:cacheprovider -k "golden or shard" > $O/pytest_marlin.log 2>&1; echo "rc=$?" >> $O/pytest_marlin.log ); tail -4 $O/pytest_marlin.log
AB_ROUNDS=2 bash tools/ab.sh marlin_amd/csrc/build_alt/libmarlin_hip_r04.so --no-seam-route > $O/ab_2p20.txt 2>&1; cut -c1-260 $O/ab_2p20.txt
AB_ROUNDS=2 bash tools/ab.sh marlin_amd/csrc/build_alt/libmarlin_hip_r04.so --no-seam-route --log-constraints 16 --pc sonic > $O/ab_2p16.txt 2>&1; cut -c1-260 $O/ab_2p16.txt
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python $OLDPWD/bench.py --steps 3 --warmup 1 \
    --no-cpu-baseline --no-seam-route --no-verify > $O/trace.log 2>&1 )
T=$(find $O/trace -name "*kernel_trace.csv" | head -1)
[ -n "$T" ] && python tools/prove_kernels.py $T > $O/last_prove_kernels_2p20.txt 2>&1 && python tools/gap_analysis.py $T 4 > $O/gaps_2p20.txt 2>&1
rm -rf $O/trace; head -16 $O/last_prove_kernels_2p20.txt
