set -x
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03e; mkdir -p $O
export TMPDIR=/tmp
timeout 120 tools/mfma_reduce_bench > $O/mfma_reduce_bench.txt 2>&1; cat $O/mfma_reduce_bench.txt
timeout 300 bash tools/fetch_calib.sh $O/fetch_calib > $O/fetch_calib.log 2>&1; tail -28 $O/fetch_calib.log; cat $O/fetch_calib/plain.log
timeout 600 python tools/ntt_dist_bench.py 8 20 22 23 24 25 > $O/ntt_dist_bench_g8.txt 2>&1; cat $O/ntt_dist_bench_g8.txt
timeout 300 python tools/ntt_dist_bench.py 2 22 24 > $O/ntt_dist_bench_g2.txt 2>&1; cat $O/ntt_dist_bench_g2.txt
timeout 300 python tools/ntt_dist_bench.py 4 22 24 > $O/ntt_dist_bench_g4.txt 2>&1; cat $O/ntt_dist_bench_g4.txt
