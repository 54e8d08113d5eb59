"""Instruction mix of the bucket-accumulation loop bodies, counted in the gfx950 ISA hipcc emits for capi.hip
(the numbers bench.py's roofline_valu uses).  Usage: python tools/loop_isa_stats.py [--curve bn254] [--json FILE] > profiles/<tag>_accum_loop_isa.txt
--json merges {"<curve>": {"fixed-base": {mad_u64, half_rate_other, full_rate}, "variable-base": {...}}} into FILE
(profiles/accum_isa_mix.json is what bench.py reads)."""
import json
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
curve = sys.argv[sys.argv.index("--curve") + 1] if "--curve" in sys.argv else "bls12_381"
json_out = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
# issue classes measured with tools/microbench (profiles/r02a_microbench_controls.txt): v_mad_u64_u32 by itself; everything else that
# multiplies, carries or is 64 bits wide issues at the same half rate; the rest at full rate
HALF = ("v_lshrrev_b64", "v_lshlrev_b64", "v_ashrrev_i64", "v_mul_lo_u32", "v_mul_hi_u32", "v_lshl_add_u64", "v_add3_u32", "v_addc_co_u32",
        "v_subb_co_u32", "v_subbrev_co_u32", "v_add_co_u32", "v_sub_co_u32", "v_mad_u32_u24", "v_mul_u32_u24")
mixes = {}
src = os.path.join(ROOT, "marlin_amd", "csrc", "capi.hip")
with tempfile.TemporaryDirectory() as td:
    out = os.path.join(td, "capi.s")
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only"] + (["-DMH_CURVE_BN254"] if curve == "bn254" else []) + [src, "-o", out],
                   check=True, stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
# rsum_kernel: the bucket reduction's row / column sums (msm_fb.cuh); its loop body is one GENERAL addition (XYZZ += XYZZ)
# accum30v_kernel: the accumulate kernel over virtual slots (round 6, the default; interleaved multiplication chains); accum30_kernel: the
# one-thread-per-bucket kernel of rounds 1-5 (MH_ACC_PARTS=0)
for kern in ("accum30v_kernel", "accum30_kernel", "accum_kernel", "rsum_kernel"):
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
    print("%s: basic blocks %s = loop body of one %s addition" % (kern, " + ".join("%s (%d)" % (n, len(i)) for n, i in hot), "general (XYZZ += XYZZ)" if kern == "rsum_kernel" else "bucket"))
    print("  instructions %d, VALU %d, v_mad_u64_u32 %d" % (len(ins), valu, c["v_mad_u64_u32"]))
    for k, v in c.most_common(14):
        print("    %-22s %5d" % (k, v))
    half = sum(v for k, v in c.items() if k.startswith("v_") and k.split("_e")[0] in HALF)
    mad = c["v_mad_u64_u32"]
    mixes[{"accum30v_kernel": "fixed-base", "accum30_kernel": "fixed-base, one thread per bucket", "accum_kernel": "variable-base", "rsum_kernel": "reduce"}[kern]] = {"mad_u64": mad, "half_rate_other": half, "full_rate": valu - mad - half}
    print("  classes: v_mad_u64_u32 %d, other half-rate %d, full-rate %d" % (mad, half, valu - mad - half))
if json_out:
    cur = json.load(open(json_out)) if os.path.exists(json_out) else {}
    cur[curve] = mixes
    cur["source"] = "tools/loop_isa_stats.py --curve <c> --json (hipcc -S of capi.hip, loop body of one bucket addition)"
    json.dump(cur, open(json_out, "w"), indent=1, sort_keys=True)
