#!/bin/bash
# Usage (GPU box, repo root): tools/clock_sample.sh <outdir> -- samples rocm-smi clocks / power while bench.py and the
# v_mad_u64_u32 microbenchmark run, to tell "issue-bound at the sustained clock" from "stalls" (DESIGN.md 4.3).
O=${1:-gpurun_out/clocks}; mkdir -p $O
sample() {  # $1 = file, runs until killed
  while true; do
    echo "t=$(date +%s.%N) $(rocm-smi --showclocks --showpower --showuse 2>/dev/null | grep -E 'sclk|mclk|Power|GPU use' | tr -s ' ' | tr '\n' ';')" >> $1
    sleep 0.05
  done
}
sample $O/bench_samples.txt & SP=$!
python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-seam-route --no-verify > $O/bench.json 2> $O/bench.err
kill $SP; wait $SP 2>/dev/null
if [ -x tools/microbench ]; then
  sample $O/microbench_samples.txt & SP=$!
  timeout 120 tools/microbench > $O/microbench.txt 2>&1
  kill $SP; wait $SP 2>/dev/null
fi
python3 - "$O" <<'PY'
import re, sys, json
o = sys.argv[1]
def stats(f):
    rows = []
    for l in open(f):
        m = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", l)
        p = re.search(r"Power \(W\): ([\d.]+)", l) or re.search(r"Socket Graphics Package Power \(W\): ([\d.]+)", l) or re.search(r"([\d.]+)\s*W", l)
        u = re.search(r"GPU use \(%\): (\d+)", l)
        if m: rows.append((int(m.group(1)), float(p.group(1)) if p else None, int(u.group(1)) if u else None))
    return rows
for name in ("bench_samples.txt", "microbench_samples.txt"):
    try:
        r = stats(o + "/" + name)
    except FileNotFoundError:
        continue
    busy = [x for x in r if x[2] is None or x[2] > 50]
    cl = sorted(x[0] for x in busy)
    print(name, "samples", len(r), "busy", len(busy), "sclk MHz min/median/max", (cl[0], cl[len(cl)//2], cl[-1]) if cl else None,
          "power W median", sorted(x[1] for x in busy if x[1] is not None)[len(busy)//2] if busy and busy[0][1] is not None else None)
PY
head -c 600 $O/bench_samples.txt
