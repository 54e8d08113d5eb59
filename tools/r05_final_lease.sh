set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=$(pwd)/gpurun_out/r05y; mkdir -p $O
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -q --durations=6 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log ); tail -5 $O/pytest.log
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python $OLDPWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-throughput --no-seam-route --no-verify > $O/trace.log 2>&1 )
T=$(find $O/trace -name "*kernel_trace.csv" | head -1)
[ -n "$T" ] && python tools/prove_kernels.py $T > $O/last_prove_kernels_2p20.txt 2>&1 && python tools/gap_analysis.py $T 4 > $O/gaps_2p20.txt 2>&1
rm -rf $O/trace; head -14 $O/last_prove_kernels_2p20.txt
bash tools/rehearse_ranks.sh r05y 20 8 4 2>&1 | tail -6
MH_MOCK_RCCL_ASYNC=1 REHEARSE_TRANSPORTS=native bash tools/rehearse_ranks.sh r05y_async 20 8 2>&1 | tail -3
