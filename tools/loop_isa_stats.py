"""Instruction mix of the bucket-accumulation loop bodies, counted in the gfx950 ISA hipcc emits for capi.hip
(the numbers bench.py's roofline_valu uses).  Usage: python tools/loop_isa_stats.py > profiles/<tag>_accum_loop_isa.txt"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "marlin_amd", "csrc", "capi.hip")
with tempfile.TemporaryDirectory() as td:
    out = os.path.join(td, "capi.s")
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", src, "-o", out],
                   check=True, stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
for kern in ("accum30_kernel", "accum_kernel"):
    starts = [i for i, l in enumerate(lines) if re.match(r"^_ZN\d+msm(fb)?\d+%s\w*:" % kern, l)]
    if len(starts) > 1:
        print("%s: %d instantiations, the first is shown" % (kern, len(starts)))
    if not starts:
        continue
    start = starts[0]
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    blocks, cur = [], None
    for l in lines[start:end]:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            cur = [m.group(1), []]
            blocks.append(cur)
            continue
        t = l.strip()
        if cur is not None and t and not t.startswith((".", ";")):
            cur[1].append(t.split()[0])
    # the loop body of one bucket addition is the set of large basic blocks of the kernel (the collision test splits it in
    # two; prologue, epilogue and the rare exact-comparison path are small)
    hot = [(n, i) for n, i in blocks if sum(1 for x in i if x.startswith("v_")) >= 400]
    ins = [x for _, i in hot for x in i]
    c = collections.Counter(ins)
    valu = sum(v for k, v in c.items() if k.startswith("v_"))
    print("%s: basic blocks %s = loop body of one bucket addition" % (kern, " + ".join("%s (%d)" % (n, len(i)) for n, i in hot)))
    print("  instructions %d, VALU %d, v_mad_u64_u32 %d" % (len(ins), valu, c["v_mad_u64_u32"]))
    for k, v in c.most_common(14):
        print("    %-22s %5d" % (k, v))
