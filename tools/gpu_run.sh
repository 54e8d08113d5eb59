#!/bin/bash
# ONE parameterised runner for a gpurun lease (replaces the 31 frozen tools/gpu_run_r03*.sh of round 3; those stay readable in
# history: `git show db7f097:tools/gpu_run_r03zz.sh`).  On the GPU box, from the repo root:
#     gpurun --timeout 1800 -- 'bash tools/gpu_run.sh <tag> <recipe> [<recipe> ...]'
# Everything a recipe writes goes to gpurun_out/<tag>/ (merged back by gpurun); what is to be judged is copied to profiles/<tag>_*.
# Recipes:
#   suite        pytest -m gpu (whole suite) + __graft_entry__.smoke()
#   bench        the default bench line (2^20, BLS12-381, MarlinKZG10, CPU baseline at the same size, seam route)
#   profile      tools/profile.sh <tag>: rocprofv3 kernel stats, accumulate dispatches, PMC traffic
#   stats        the first third of `profile`: rocprofv3 --kernel-trace --stats of a short bench + the accumulate dispatches (no PMC passes)
#   sq           tools/profile_sq.sh <tag>: SQ wave-cycle breakdown of the accumulate kernel
#   sims         one rank of 2 / 4 / 8 simulated on this GPU at 2^20, one of 8 at 2^22 (exchanges: Python callbacks making local copies)
#   sims_native  the same through the native transport (local copies issued from C++ by the solo stand-in for librccl)
#   configs      the other BASELINE configurations (2^16 Sonic, 2^18, 2^22, BLS Sonic, BN254 Marlin / Sonic)
#   small        the reference's own bench shape (2^16, SonicKZG10) with a kernel trace of the last prove and its gap analysis
#   seam         the seam route (host pointers), with and without the short uploads of mh_ntt_len
#   check        tests/test_gpu_check.py (the MH_CHECK invariants and their fault injection)
#   soak=N[:D[:s]]  tools/soak_sliced.py: N iterations of the 8-process sliced-MSM scenario at MH_CHECK=2, MH_DIAG=D, s = AMD_SERIALIZE_KERNEL=3
#   ab=<lib.so>  tools/ab.sh: alternate the in-tree build and another build of the same ABI
set -u
TAG=${1:?tag}; shift
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=$(pwd)/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
B="timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-throughput --no-seam-route"
line() { python - "$@" <<'PY'
import json, sys
for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        p = d.get("proof") or {}
        print(f.split("/")[-1], d["ms_per_step"], d["value"], {k: v for k, v in d["breakdown_ms_per_step"].items() if k != "measured_on"},
              "verified", p.get("verified"), "golden", (p.get("oracle_golden") or {}).get("byte_identical"))
    except Exception as e:
        print(f, "ERR", e)
PY
}
for R in "$@"; do
  case $R in
    suite)
      ( timeout 2400 python -m pytest tests -m gpu -q --durations=45 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log ); tail -8 $O/pytest.log
      python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.txt ;;
    bench)
      timeout 1200 python bench.py > $O/bench_default.json 2> $O/bench_default.err; line $O/bench_default.json ;;
    profile)
      timeout 1500 bash tools/profile.sh $TAG > $O/profile.log 2>&1; tail -3 $O/profile.log ;;
    stats)
      PROFILE_STATS_ONLY=1 timeout 700 bash tools/profile.sh $TAG > $O/stats.log 2>&1; cp gpurun_out/prof_$TAG/*.json gpurun_out/prof_$TAG/*.csv $O/ 2>/dev/null; tail -3 $O/stats.log ;;
    sq)
      timeout 1200 bash tools/profile_sq.sh $TAG --no-seam-route > $O/sq.log 2>&1; cp gpurun_out/prof_$TAG/sq_counters.json $O/ 2>/dev/null; tail -25 $O/sq.log ;;
    sims)
      $B --simulate-rank 1/2 > $O/sim_1_2.json 2>/dev/null; $B --simulate-rank 3/4 > $O/sim_3_4.json 2>/dev/null
      $B --simulate-rank 5/8 > $O/sim_5_8.json 2>/dev/null; $B --log-constraints 22 --simulate-rank 3/8 > $O/sim_3_8_2p22.json 2>/dev/null
      line $O/sim_*.json ;;
    sims_native)    # the same ranks through the native transport's own code path (solo stand-in: local copies issued from C++)
      $B --transport native --simulate-rank 1/2 > $O/simn_1_2.json 2>/dev/null; $B --transport native --simulate-rank 3/4 > $O/simn_3_4.json 2>/dev/null
      $B --transport native --simulate-rank 5/8 > $O/simn_5_8.json 2>/dev/null; $B --transport native --log-constraints 22 --simulate-rank 3/8 > $O/simn_3_8_2p22.json 2>/dev/null
      line $O/simn_*.json ;;
    configs)
      $B --log-constraints 16 --pc sonic --steps 20 > $O/bench_reference_shape_2p16_sonickzg10.json 2>/dev/null
      $B --log-constraints 18 > $O/bench_2p18.json 2>/dev/null
      $B --log-constraints 22 --steps 5 --warmup 2 > $O/bench_2p22.json 2>/dev/null
      $B --pc sonic > $O/bench_bls12_381_sonickzg10_2p20.json 2>/dev/null
      MARLIN_AMD_CURVE=bn254 $B > $O/bench_bn254_marlinkzg10_2p20.json 2>/dev/null
      MARLIN_AMD_CURVE=bn254 $B --pc sonic > $O/bench_bn254_sonickzg10_2p20.json 2>/dev/null
      line $O/bench_reference_shape_*.json $O/bench_2p18.json $O/bench_2p22.json $O/bench_bls12_381_sonic*.json $O/bench_bn254_*.json ;;
    small)
      $B --log-constraints 16 --pc sonic --steps 20 > $O/bench_2p16_sonic.json 2>/dev/null; line $O/bench_2p16_sonic.json
      ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace_2p16 -o t -- python $OLDPWD/bench.py --steps 3 --warmup 1 \
          --no-cpu-baseline --no-throughput --no-seam-route --no-verify --log-constraints 16 --pc sonic > $O/trace_2p16.log 2>&1 )
      T=$(find $O/trace_2p16 -name "*kernel_trace.csv" | head -1)
      [ -n "$T" ] && python tools/gap_analysis.py $T 4 > $O/gaps_2p16_sonic.txt 2>&1 && python tools/prove_kernels.py $T > $O/last_prove_kernels_2p16_sonic.txt 2>&1
      rm -rf $O/trace_2p16; tail -30 $O/gaps_2p16_sonic.txt ;;
    seam)
      timeout 600 python bench.py --workload seam-route --steps 3 --warmup 1 > $O/bench_seam.json 2> $O/bench_seam.err
      BENCH_SEAM_FULL_UPLOAD=1 timeout 600 python bench.py --workload seam-route --steps 3 --warmup 1 > $O/bench_seam_full_upload.json 2>> $O/bench_seam.err
      line $O/bench_seam.json $O/bench_seam_full_upload.json ;;
    suite_check)    # the whole GPU suite with the pipeline's invariants on in every process (MH_CHECK=1)
      ( MH_CHECK=1 timeout 2400 python -m pytest tests -m gpu -q --durations=10 -p no:cacheprovider > $O/pytest_mh_check_1.log 2>&1; echo "pytest rc=$?" >> $O/pytest_mh_check_1.log ); tail -8 $O/pytest_mh_check_1.log ;;
    new)            # the tests of this round's new code, first
      ( timeout 1500 python -m pytest tests/test_gpu_contexts.py tests/test_gpu_check.py tests/test_capi_symbols.py tests/test_gpu_poisoned_allocations.py tests/test_gpu_reentrancy.py -m gpu -q -x --durations=20 -p no:cacheprovider > $O/pytest_new.log 2>&1; echo "pytest rc=$?" >> $O/pytest_new.log
        timeout 900 python -m pytest tests/test_gpu_rccl_native.py tests/test_gpu_msm.py -m gpu -q -x -k "fails_mid_prove or selftest or native_rccl_world_1 or skewed" --durations=10 -p no:cacheprovider >> $O/pytest_new.log 2>&1; echo "pytest rc=$?" >> $O/pytest_new.log ); grep -E "passed|failed|rc=|Error|error" $O/pytest_new.log | tail -30 ;;
    simtrace)       # kernel trace of one simulated rank of 8 at 2^20: where a rank's time goes (tools/prove_kernels.py on the last prove)
      ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace_sim58 -o t -- python $OLDPWD/bench.py --steps 3 --warmup 1 \
          --no-cpu-baseline --no-seam-route --no-verify --simulate-rank 5/8 > $O/trace_sim58.log 2>&1 )
      T=$(find $O/trace_sim58 -name "*kernel_trace.csv" | head -1)
      [ -n "$T" ] && python tools/prove_kernels.py $T > $O/sim_5_8_last_prove_kernels.txt 2>&1 && python tools/gap_analysis.py $T 4 > $O/sim_5_8_gaps.txt 2>&1
      [ -n "$T" ] && python tools/prove_timeline.py $T 4 > $O/sim_5_8_timeline.txt 2>&1
      rm -rf $O/trace_sim58; head -70 $O/sim_5_8_last_prove_kernels.txt ;;
    one=*)          # one=<pytest node id or -k expression file::test>: a single test, verbose
      ( timeout 900 python -m pytest "${R#one=}" -m gpu -x -q -p no:cacheprovider > $O/pytest_one.log 2>&1; echo "pytest rc=$?" >> $O/pytest_one.log ); tail -60 $O/pytest_one.log ;;
    hosttrace)      # HIP API + kernel trace of the timed proves at 2^20 and of one simulated rank of 8: where the HOST spends its time
      for cfg in "2p20:" "sim58:--simulate-rank 5/8"; do
        name=${cfg%%:*}; extra=${cfg#*:}
        ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --hip-runtime-trace --output-format csv -d $O/ht_$name -o t -- python $OLDPWD/bench.py --steps 3 --warmup 1 \
            --no-cpu-baseline --no-seam-route --no-verify $extra > $O/ht_$name.log 2>&1 )
        A=$(find $O/ht_$name -name "*hip_api_trace.csv" | head -1); T=$(find $O/ht_$name -name "*kernel_trace.csv" | head -1)
        [ -n "$A" ] && [ -n "$T" ] && python tools/host_gaps.py $A $T 4 12 > $O/host_gaps_$name.txt 2>&1
        rm -rf $O/ht_$name; tail -45 $O/host_gaps_$name.txt
      done ;;
    msmtests)       # the MSM / check / golden-proof tests (after a change to the fixed-base pipeline), invariants on
      ( MH_CHECK=2 timeout 1500 python -m pytest tests/test_gpu_msm.py tests/test_gpu_check.py tests/test_gpu_dist_blocks.py tests/test_gpu_reentrancy.py -m gpu -q -x --durations=8 -p no:cacheprovider > $O/pytest_msm.log 2>&1; echo "pytest rc=$?" >> $O/pytest_msm.log
        MH_CHECK=1 timeout 1500 python -m pytest tests/test_gpu_marlin.py tests/test_gpu_parity_pins.py -m gpu -q -x -k "golden or sharded or zero_matrix or sonic_proof or full_size" --durations=8 -p no:cacheprovider >> $O/pytest_msm.log 2>&1; echo "pytest rc=$?" >> $O/pytest_msm.log ); grep -E "passed|failed|rc=|Error|error|assert" $O/pytest_msm.log | tail -30 ;;
    trace20)        # kernel trace of the timed proves at 2^20: the last prove's kernels and the idle time between dispatches
      ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace_2p20 -o t -- python $OLDPWD/bench.py --steps 3 --warmup 1 \
          --no-cpu-baseline --no-seam-route --no-verify > $O/trace_2p20.log 2>&1 )
      T=$(find $O/trace_2p20 -name "*kernel_trace.csv" | head -1)
      [ -n "$T" ] && python tools/prove_kernels.py $T > $O/last_prove_kernels_2p20.txt 2>&1 && python tools/gap_analysis.py $T 4 > $O/gaps_2p20.txt 2>&1
      [ -n "$T" ] && python tools/prove_timeline.py $T 4 > $O/timeline_2p20.txt 2>&1
      rm -rf $O/trace_2p20; head -40 $O/last_prove_kernels_2p20.txt; head -8 $O/gaps_2p20.txt ;;
    sweep=*)        # sweep=VAR:v1,v2,...: the default bench at 2^20 under each value of one environment variable (same box, back to back, twice)
      IFS=: read -r VAR VALS <<< "${R#sweep=}"
      for rep in 1 2; do for V in ${VALS//,/ }; do
        env $VAR=$V $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=d['breakdown_ms_per_step']; print('$VAR=$V', d['ms_per_step'], 'accum', b['msm_accum'], 'sort+reduce', b['msm_sort_and_reduce_stages'])"
      done; done | tee -a $O/sweep_$VAR.txt ;;
    sweepsim=*)     # sweepsim=VAR:v1,v2,...: one simulated rank of 8 at 2^20 under each value of one environment variable (twice; "-" = unset)
      IFS=: read -r VAR VALS <<< "${R#sweepsim=}"
      for rep in 1 2; do for V in ${VALS//,/ }; do
        ( if [ "$V" = "-" ]; then unset $VAR; else export $VAR=$V; fi
          $B --simulate-rank 5/8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=d['breakdown_ms_per_step']; print('$VAR=$V', d['ms_per_step'], 'accum', b['msm_accum'], 'sort+reduce', b['msm_sort_and_reduce_stages'], 'glue', b['glue'], 'host', b['host_and_other'])" )
      done; done | tee -a $O/sweepsim_$VAR.txt ;;
    check)          # the MH_CHECK tests by themselves
      ( timeout 900 python -m pytest tests/test_gpu_check.py -m gpu -q -x --durations=8 -p no:cacheprovider > $O/pytest_check.log 2>&1; echo "pytest rc=$?" >> $O/pytest_check.log ); tail -25 $O/pytest_check.log ;;
    soak=*)         # tools/soak_sliced.py: soak=<iterations>[:<MH_DIAG>[:serialize]] -- 8 processes on the GPU, MH_CHECK=2
      IFS=: read -r IT DG SER <<< "${R#soak=}"
      S=$O/soak_${IT}_diag${DG:-0}${SER:+_serialized}; mkdir -p $S
      ( export MH_DIAG=${DG:-0}; [ -n "${SER:-}" ] && export AMD_SERIALIZE_KERNEL=3
        timeout 3000 python tools/soak_sliced.py --world 8 --iters $IT --check 2 --sharded-c ${SOAK_SHARDED_C:-16} --out $S --timeout 2900 > $S/soak.log 2>&1; echo "soak rc=$?" >> $S/soak.log )
      grep -E "^SOAK|soak rc|soak:" $S/soak.log | tail -12 ;;
    fresh=*)        # fresh=<launches>: the soak's scenario from a cold start, again and again (a process's FIRST batch runs on fresh allocations)
      NL=${R#fresh=}; S=$O/fresh_$NL; mkdir -p $S; : > $S/fresh.log
      for i in $(seq 1 $NL); do timeout 300 python tools/soak_sliced.py --world 8 --iters 3 --check 2 --out $S --port $((29900 + i % 50)) --timeout 200 2>&1 | grep -E "^SOAK" >> $S/fresh.log; done
      python - $S/fresh.log <<'PY'
import json, sys
rows = [json.loads(l[5:]) for l in open(sys.argv[1]) if l.startswith("SOAK ")]
print("FRESH launches %d, sliced MSMs %d, mismatches %d, errors %d, violations %d" % (len(rows), sum(r["sliced_msms"] for r in rows),
      sum(r["mismatches"] for r in rows), sum(r["errors"] for r in rows), sum(r["violations"] for r in rows)))
PY
      ;;
    absim=*)        # absim=<lib.so>: the same alternation for one simulated rank of 8 at 2^20
      bash tools/ab.sh "${R#absim=}" --no-seam-route --simulate-rank 5/8 > $O/ab_sim_5_8.txt 2>&1; cut -c1-330 $O/ab_sim_5_8.txt ;;
    abcfg=*)        # abcfg=<lib.so>: the in-tree build and another build of the same ABI on the other BASELINE configurations, back to back
      OTHER=${R#abcfg=}
      for cfg in "--log-constraints 16 --pc sonic --steps 20" "--log-constraints 18" "--log-constraints 22 --steps 5 --warmup 2" "--pc sonic"; do
        for v in new old; do
          ( if [ $v = old ]; then export MARLIN_AMD_LIB=$OTHER; fi
            $B $cfg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=d['breakdown_ms_per_step']; print('$v', '$cfg', d['ms_per_step'], 'accum', b['msm_accum'], 'sort+reduce', b['msm_sort_and_reduce_stages'], 'golden', ((d.get('proof') or {}).get('oracle_golden') or {}).get('byte_identical'))" )
        done
      done | tee $O/ab_configs.txt ;;
    ab=*)
      bash tools/ab.sh "${R#ab=}" --no-seam-route > $O/ab.txt 2>&1; cut -c1-330 $O/ab.txt ;;
    *) echo "unknown recipe $R" ;;
  esac
done
