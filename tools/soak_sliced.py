#!/usr/bin/env python3
"""Soak of the sliced MSM under multi-process load on ONE GPU, with the pipeline's invariants on (MH_CHECK, msm_check.cuh).

VERDICT r05 item 1: `tests/test_gpu_dist_blocks.py::test_distributed_ntt_and_sliced_msm[8-logs2]` once reported "rank 0: sliced MSM
differs" with 8 processes sharing the GPU and was never seen again.  This runs that exact scenario -- 8 ranks, gloo, the same SRS,
window width and job shapes -- in a loop with fresh scalars every iteration and, per iteration and rank, compares

  * the sliced result (mh_msm_batch_sliced_dev, partial points all-gathered over gloo and summed)  with the unsliced MSM
    (mh_msm_batch_dev) of the whole vectors -- the comparison that failed;
  * this rank's OWN partial points (combine = 0) with the variable-base path on the gathered bases of its slice -- an MSM that
    shares no kernel with the fixed-base path after the scalars are read.

On the first mismatch every rank writes what it has to <out>/soak_rank<r>.json: the iteration, which jobs differ, its partial
points, the reference's, whether the unsliced result reproduces, and the stage-by-stage text of the library's last checked batch
(mh_check_report).  A violated invariant makes the call itself fail with MH_ECHECK naming the stage; that is recorded the same way.

Round 6's accumulate kernel (virtual slots; in a bucket-range shard the long buckets are accumulated in parts and merged) gets its
own leg: --sharded-c C builds a second window table of width C over the same points and, per iteration, runs the three MSMs sharded by
BUCKET RANGE over the ranks (mh_msm_batch_sharded_dev: every rank's partial sums all-gathered and added) against the unsliced MSM.

  python tools/soak_sliced.py --world 8 --iters 900 --check 2 --out gpurun_out/soak     # ~5,400 sliced MSMs per run
  MH_DIAG=1 ... / MH_DIAG=2 ... / AMD_SERIALIZE_KERNEL=3 ...                            # the bisection configurations
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(args):
    sys.path.insert(0, ROOT)
    import ctypes as C
    import numpy as np
    import torch.distributed as dist
    import marlin_amd as M
    from marlin_amd import dist as MD, _lib
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    M.init(0)
    lib = _lib.load()
    _lib.check(lib.mh_check_level(args.check), "mh_check_level")
    MD.enable_sharded_prove(dist)
    MD.enable_alltoall(dist)

    def report():
        buf = C.create_string_buffer(8192)
        cnt = (C.c_uint64 * 2)()
        lib.mh_check_report(buf, 8192, cnt)
        return {"text": buf.value.decode(errors="replace"), "batches": int(cnt[0]), "violations": int(cnt[1])}

    def rand_fr(rng, n):
        x = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
        x[:, 3] &= np.uint64((1 << 59) - 1)
        return x

    aff = lambda a: [tuple(int(v) for v in M.g1_to_affine(r)[0]) for r in a]
    hexpt = lambda a: ["".join("%016x" % int(v) for v in r) for r in a]
    out = {"rank": rank, "world": world, "check": args.check, "mismatches": [], "errors": [], "iters": 0, "sliced_msms": 0,
           "env": {k: os.environ.get(k) for k in ("MH_DIAG", "AMD_SERIALIZE_KERNEL", "MH_CHECK")}}

    def dump():
        out["report"] = report()
        with open(os.path.join(args.out, "soak_rank%d.json" % rank), "w") as f:
            json.dump(out, f, indent=1)

    # the distributed transforms of the original scenario run first (they size the exchange buffers the way the test did)
    rng0 = np.random.default_rng(11)
    for log_n in args.ntt_logs:
        n = 1 << log_n
        x = rand_fr(rng0, n)
        want = M.ntt(x)
        mine = MD.c_layout_slice(x, rank, world)
        d_in, d_out = M.DeviceBuffer.from_numpy(mine), M.DeviceBuffer(32 * (n // world))
        MD.ntt_dist_dev(d_in, d_out, log_n)
        idx = MD.m_layout_indices(n, rank, world)
        if not np.array_equal(d_out.download((n // world, 4)), want[idx]):
            out["errors"].append({"where": "ntt_dist 2^%d" % log_n})
        d_in.free(); d_out.free()

    n = 1 << args.msm_log
    tau = np.array([0x1234567, 0, 0, 0], dtype=np.uint64)
    B = M.Bases.srs_powers(tau, n + 64)
    B.precompute(args.c)
    allb = B.download()
    B2 = None
    if args.sharded_c:
        B2 = M.Bases(allb)
        B2.precompute(args.sharded_c)
    out["bucket_range_sharded_msms"] = 0
    n2 = n - 5
    len1, len2 = len(range(rank, n, world)), len(range(rank, n2, world))
    # the bases of this rank's three slices as contiguous sets WITHOUT a table: the variable-base path serves them
    refsets = [M.Bases(np.ascontiguousarray(allb[first:first + world * ln:world][:ln])) for first, ln in ((rank, len1), (37 + rank, len2), (rank, len2))]
    d1, d2 = M.DeviceBuffer(32 * n), M.DeviceBuffer(32 * n2)
    e1, e2 = M.DeviceBuffer(32 * len1), M.DeviceBuffer(32 * len2)
    t0 = time.time()
    for it in range(args.iters):
        rng = np.random.default_rng(1000 + it)                   # the same vectors on every rank
        s1, s2 = rand_fr(rng, n), rand_fr(rng, n2)
        l1, l2 = MD.c_layout_slice(s1, rank, world), MD.c_layout_slice(s2, rank, world)
        d1.upload(s1); d2.upload(s2); e1.upload(l1); e2.upload(l2)
        jobs = [(rank, e1, len1), (37 + rank, e2, len2), (rank, e2, len2)]
        try:
            whole = M.msm_batch_dev([(B, 0, d1, n), (B, 37, d2, n2), (B, 0, d2, n2)])
            got = MD.msm_batch_sliced_dev(B, jobs, world)
            mine = MD.msm_batch_sliced_dev(B, jobs, world, combine=False)
            refs = [M.msm(refsets[0], l1), M.msm(refsets[1], l2), M.msm(refsets[2], l2)]
            shd = M.msm_batch_sharded_dev([(B2, 0, d1, n), (B2, 37, d2, n2), (B2, 0, d2, n2)]) if B2 is not None else None
        except _lib.MarlinHipError as e:
            out["errors"].append({"iter": it, "error": str(e)})
            dump()
            break
        out["iters"] = it + 1
        out["sliced_msms"] += 6
        if shd is not None:
            out["bucket_range_sharded_msms"] += 3
            bad_shd = [j for j in range(3) if aff(shd)[j] != aff(whole)[j]]
            if bad_shd:
                out["mismatches"].append({"iter": it, "bucket_range_sharded_differs_in_jobs": bad_shd, "sharded": hexpt(shd), "whole": hexpt(whole)})
                dump()
        bad_sum = [j for j in range(3) if aff(got)[j] != aff(whole)[j]]
        bad_own = [j for j in range(3) if aff(mine)[j] != aff(refs)[j]]
        if bad_sum or bad_own:
            again = M.msm_batch_dev([(B, 0, d1, n), (B, 37, d2, n2), (B, 0, d2, n2)])
            mine2 = MD.msm_batch_sliced_dev(B, jobs, world, combine=False)
            out["mismatches"].append({"iter": it, "sum_differs_in_jobs": bad_sum, "own_partial_differs_in_jobs": bad_own,
                                      "unsliced_reproduces": aff(again) == aff(whole), "own_partial_reproduces": aff(mine2) == aff(mine),
                                      "own_partial_second_run_right": aff(mine2) == aff(refs),
                                      "got": hexpt(got), "whole": hexpt(whole), "mine": hexpt(mine), "refs": hexpt(refs)})
            dump()
            if len(out["mismatches"]) >= 5:
                break
        if rank == 0 and (it + 1) % 100 == 0:
            print("soak: %d iterations, %.1f s, %d mismatches" % (it + 1, time.time() - t0, len(out["mismatches"])), flush=True)
    out["seconds"] = time.time() - t0
    dump()
    print("rank %d done: %d iterations, %d sliced MSMs, %d mismatches, %d errors, %d batches checked, %d violations"
          % (rank, out["iters"], out["sliced_msms"], len(out["mismatches"]), len(out["errors"]), out["report"]["batches"], out["report"]["violations"]), flush=True)
    try:
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        pass
    return 1 if (out["mismatches"] or out["errors"]) else 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--iters", type=int, default=900)
    ap.add_argument("--check", type=int, default=2)
    ap.add_argument("--msm-log", type=int, default=15)
    ap.add_argument("--c", type=int, default=14)
    ap.add_argument("--sharded-c", type=int, default=0, help="also run the MSMs sharded by bucket range over a second table of this window width (16: 16 partitions, cuts at 8 ranks)")
    ap.add_argument("--ntt-logs", type=int, nargs="*", default=[6, 7, 13, 20])
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "soak"))
    ap.add_argument("--port", type=int, default=29871)
    ap.add_argument("--timeout", type=int, default=3000)
    ap.add_argument("--worker", action="store_true")
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    if args.worker:
        sys.exit(worker(args))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(args.port), WORLD_SIZE=str(args.world))
    cmd = [sys.executable, os.path.abspath(__file__), "--worker"] + [a for a in sys.argv[1:]]
    procs = [subprocess.Popen(cmd, env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(args.world)]
    rc = 0
    deadline = time.time() + args.timeout
    for r, p in enumerate(procs):
        try:
            so, _ = p.communicate(timeout=max(1, deadline - time.time()))
        except subprocess.TimeoutExpired:
            p.kill()
            so, _ = p.communicate()
            so += "\n[killed: timeout]"
        print("---- rank %d (exit %s)\n%s" % (r, p.returncode, so[-3000:]))
        rc = rc or (p.returncode or 0)
    # one summary line for the whole run
    tot = {"world": args.world, "iters": 0, "sliced_msms": 0, "bucket_range_sharded_msms": 0, "mismatches": 0, "errors": 0, "batches_checked": 0, "violations": 0}
    for r in range(args.world):
        try:
            d = json.load(open(os.path.join(args.out, "soak_rank%d.json" % r)))
        except Exception:
            rc = rc or 1
            continue
        tot["iters"] = max(tot["iters"], d["iters"]); tot["sliced_msms"] += d["sliced_msms"]; tot["bucket_range_sharded_msms"] += d.get("bucket_range_sharded_msms", 0)
        tot["mismatches"] += len(d["mismatches"]); tot["errors"] += len(d["errors"])
        tot["batches_checked"] += d["report"]["batches"]; tot["violations"] += d["report"]["violations"]
    tot["env"] = {k: os.environ.get(k) for k in ("MH_DIAG", "AMD_SERIALIZE_KERNEL")}
    tot["check"] = args.check
    print("SOAK " + json.dumps(tot))
    with open(os.path.join(args.out, "soak_summary.json"), "w") as f:
        json.dump(tot, f)
    sys.exit(1 if rc else 0)


if __name__ == "__main__":
    main()
