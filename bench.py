#!/usr/bin/env python3
"""bench.py -- Marlin prover hot path on MI355X.

Metric (BASELINE.json): R1CS constraints / second for Marlin::prove on BLS12-381
at 2^20 constraints (DummyCircuit of /root/reference benches/bench.rs:26-66,
MarlinKZG10).  One "step" = one pass of the prove hot path over one instance:

  workload "marlin-prove" (default): one full Marlin::prove (mh_marlin_prove_dev -- formatted input and witness
      already in HBM, as the measurement contract prescribes; BENCH_HOST_INPUTS=1 times the host-pointer entry point
      mh_marlin_prove, the PCIe-inclusive figure --: AHP rounds, Fiat-Shamir,
      commitments, batch opening -- 16 NTTs and 13 large MSMs do the work of the reference's 30 and 15,
      DESIGN.md section 2) of DummyCircuit with its index and SRS resident in HBM; the proof it emits is
      byte-identical to the oracle's at the sizes the oracle reaches (tests/test_gpu_marlin.py) and
      verifies at 2^20.
  workload "hotpath-inventory": the 30 NTTs and 15 large MSMs of one prove on synthetic vectors
      (kernel-only view, --workload hotpath-inventory).

Contract: `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line on
rank 0.  N > 1 runs one rank per GPU: either the caller launches them (`python -m torch.distributed.run
--nproc-per-node N ... bench.py --gpus N`: RANK / WORLD_SIZE are in the environment) or, when bench.py is started
plainly with --gpus N > 1, it re-executes itself under torch.distributed.run with N ranks on 127.0.0.1.
Every rank runs the whole prover but multiplies only its share of each MSM; the 144-byte partial points
are exchanged with an RCCL all_gather and added on every rank, so all ranks derive the
same transcript (strong scaling).

The timed region is `Marlin::prove` from the padded R1CS instance + witness on: constraint synthesis
(`generate_constraints` with construct_matrices, src/ahp/prover.rs:217-230 -- ark-relations host code, out of scope per
SURVEY.md 2.2 E6) is NOT included; config.workload says so.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# VALU issue rates measured on MI355X with tools/microbench (profiles/r02a_microbench_controls.txt), T lane-ops/s over the
# whole chip: plain 32-bit ops (v_add_u32, v_and_b32, v_lshrrev_b32; v_fma_f32 is the same) issue at ~63, everything that
# multiplies, carries or is 64 bits wide at about half of that -- v_mad_u64_u32 33.0, v_mul_lo_u32 / v_add3_u32 /
# v_lshrrev_b64 / v_lshl_add_u64 / v_addc_co_u32 35-36.  (The guide's 2-cycle wave64 issue is the 63 T/s class.)
VALU_CLASS_RATE_T = {"mad_u64": 33.0, "half_rate_other": 35.5, "full_rate": 63.1}
# instruction mix of ONE bucket addition (XYZZ += affine table point: 8 M + 2 S, Y3 as two products under one reduction,
# on 13 x 30-bit limbs) in the loop body of msmfb::accum30v_kernel, counted in the ISA (tools/loop_isa_stats.py ->
# profiles/r02b_accum_loop_isa.txt): 4214 VALU = 3055 v_mad_u64_u32 + 497 other half-rate (274 v_lshrrev_b64, 118
# v_mul_lo_u32, 58 v_lshl_add_u64, 47 v_add3_u32) + 662 full-rate; the variable-base kernel (32-bit limbs): 7839 VALU
# = 2880 mad + 2880 addc + 2079 others.
ACCUM_MIX = {"fixed-base": {"mad_u64": 3055, "half_rate_other": 497, "full_rate": 662},
             "variable-base": {"mad_u64": 2880, "half_rate_other": 2880 + 445 + 120, "full_rate": 1514}}


def accum_mix(curve):
    """The mix of the library that is loaded: profiles/accum_isa_mix.json holds the ISA count per curve (BN254's Fq has 9 limbs
    of 30 bits, BLS12-381's 13 -- a bucket addition is 1874 VALU instructions there, not 4214); the BLS12-381 constants above
    are the fallback when the file is absent."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "accum_isa_mix.json")))[curve]
    except (OSError, KeyError, ValueError):
        return ACCUM_MIX if curve == "bls12_381" else None


def valu_bound_adds_per_s(mix):
    """bucket additions per second the chip could issue if nothing but this instruction mix ever stalled."""
    return 1.0 / sum(cnt / (VALU_CLASS_RATE_T[k] * 1e12) for k, cnt in mix.items())


def rand_fr_np(rng, n):
    """n pseudo-random Montgomery-form Fr elements (< 2^254 < r) as (n,4) uint64."""
    x = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
    x[:, 3] &= np.uint64((1 << 61) - 1)
    return x


class HotPathInventory:
    """NTT + MSM inventory of one Marlin::prove at N constraints, sharded over `world` ranks."""

    def __init__(self, M, log_n, rank, world, seed=1):
        from marlin_amd import workload as W
        self.M = M
        self.N = 1 << log_n
        self.rank, self.world = rank, world
        H, K = self.N, 4 * self.N
        self.ntts = W.ntt_inventory(H, K)
        self.msms, self.small = W.msm_inventory(H, K)
        self.alg_ntt_bytes, self.alg_msm_bytes = W.algorithmic_bytes(H, K)
        rng = np.random.default_rng(seed + rank)
        # --- SRS shard: powers tau^i for i in this rank's slice of [0, 4N) -------------
        tau = np.array([0x9c5d7e2b4a6f8091, 0x1f3a, 0, 0], dtype=np.uint64)   # any non-zero Montgomery value
        self.srs_n = K
        self.lo = (K * rank) // world
        self.hi = (K * (rank + 1)) // world
        self.bases = M.Bases.srs_powers(tau, self.hi - self.lo, first=self.lo)
        # --- polynomial buffers (device resident) ----------------------------------------
        max_ntt = max(lg for lg, _, _ in self.ntts)
        self.buf_a = M.DeviceBuffer.from_numpy(rand_fr_np(rng, 1 << max_ntt))
        self.buf_b = M.DeviceBuffer(32 << max_ntt)
        # scalars for this rank's shard of the largest MSM
        self.scal = M.DeviceBuffer.from_numpy(rand_fr_np(rng, self.hi - self.lo))

    def step(self, dist=None, torch=None):
        M = self.M
        for lg, inverse, _ in self.ntts:
            M.ntt_dev(self.buf_a, self.buf_b, lg, inverse=inverse)
        partials = []
        for n, _ in self.msms:
            # every rank takes an equal share; its bases are the first (hi-lo) points of its shard
            cnt = (n * (self.rank + 1)) // self.world - (n * self.rank) // self.world
            cnt = min(cnt, self.bases.n)
            partials.append(M.msm_dev(self.bases, self.scal, cnt, base_offset=0, montgomery=True))
        if self.world > 1:
            from marlin_amd import dist as MD
            dev = torch.device("cuda") if dist.get_backend() == "nccl" else None
            partials = MD.combine_partials(MD.allgather_partials(np.stack(partials), dist, dev))
        return partials


class MarlinProve:
    """One Marlin::prove of DummyCircuit (benches/bench.rs:26-66) at 2^log_n constraints."""

    def __init__(self, M, log_n, pc="marlin"):
        from marlin_amd import marlin as GM, workload as W
        self.M, self.GM = M, GM
        self.N = 1 << log_n
        n = self.N
        self.alg_ntt_bytes, self.alg_msm_bytes = W.algorithmic_bytes(n, 4 * n)
        self.msms, _ = W.msm_inventory(n, 4 * n)
        tau, gamma = 0x1f3a9c5d7e2b4a6f8091a2b3c4d5e6f708192a3b4c5d6e7f, 0x5eed5eed5eed5eed0123456789abcdef
        # the witness values, SRS trapdoors and zk seed of tests/golden/marlin_proofs_xl.json (the first two draws of the
        # reference's test_rng): at the sizes that file holds, the proof this bench makes IS the oracle's golden proof
        a, b = 0x674e1d7463d34c49f9c9f388646067d796542ccbf66f38d3ab574d0ee422c588, 0x5fb51e0ee491c6f26f2fd3ab01162c4d3ad3aff73fc213510ebbf34faa74c07e
        try:        # the other curve draws other values from the same rng: take them from that configuration's golden file
            from marlin_amd import _lib as _L0
            g = json.load(open(os.path.join(ROOT, "tests", "golden", "marlin_proofs_xl_%s_%s.json" % (_L0.CURVE, pc))))
            a, b = int(g["cases"][0]["a"], 16), int(g["cases"][0]["b"], 16)
        except (OSError, KeyError, IndexError, ValueError):
            pass
        self.a, self.b = a, b
        t0 = time.time()
        self.tau, self.gamma = tau, gamma
        self.srs = GM.universal_setup(n, n, 3 * n, tau, gamma, pc=pc)
        nc, ni, mats, self.inst, self.wit = GM.dummy_circuit(a, b, 10, n)
        self.pk = GM.index(self.srs, nc, ni, mats, pc=pc)
        self.setup_s = time.time() - t0
        self.seed = bytes(range(32))
        self.proof = None
        # inputs resident in HBM when the timed region starts: the formatted input and the witness are uploaded once
        # (mh_marlin_prove_dev); the host-pointer entry point adds 32 B per constraint of PCIe copy
        self.d_inst = M.DeviceBuffer.from_numpy(np.ascontiguousarray(self.inst))
        self.d_wit = M.DeviceBuffer.from_numpy(np.ascontiguousarray(self.wit))
        self.host_inputs = bool(os.environ.get("BENCH_HOST_INPUTS"))

    def step(self, dist=None, torch=None):
        if self.host_inputs:
            self.proof = self.GM.prove(self.pk, self.inst, self.wit, self.seed)
        else:
            self.proof = self.GM.prove_dev(self.pk, self.d_inst, self.d_wit, self.seed)
        return self.proof


class SeamRoute:
    """The route BASELINE.json's north_star describes literally: the reference's Rust host code unchanged, every
    `GeneralEvaluationDomain::{fft, ifft}` going to mh_ntt (seam B2, src/ahp/prover.rs:280-287 and the 30 call sites of
    SURVEY.md Appendix A) and every `VariableBaseMSM::multi_scalar_mul` under PC::commit / PC::open_combinations
    (src/lib.rs:172,193,213,292) to mh_msm (seam B1) -- with HOST pointers, because the Rust prover keeps its
    polynomials in `Vec<Fr>`.  One step = the 30 transforms and the 15 large MSMs of one prove on host vectors of the
    right sizes (pageable memory, like a Vec), against the SRS resident on the device with its window table built once
    (PC::trim).  Everything else the Rust prover does on the host between those calls is NOT included: this is a lower
    bound of what the seam route costs per proof, to be read beside the device-resident prover's number."""

    def __init__(self, M, log_n, bases=None, seed=3, batched=True):
        from marlin_amd import workload as W, _lib
        self.M, self.lib, self.curve = M, _lib.load(), _lib.CURVE_ID
        self.batched = batched
        self.N = 1 << log_n
        H, K = self.N, 4 * self.N
        self.ntts = W.ntt_inventory(H, K)
        # what each Vec holds when the patched ark-poly hands it over, before its zero-padding: only that much is uploaded
        # (mh_ntt_len); BENCH_SEAM_FULL_UPLOAD=1 uploads whole domains (the round-3 measurement)
        self.in_lens = [1 << lg for lg, _, _ in self.ntts] if os.environ.get("BENCH_SEAM_FULL_UPLOAD") else W.ntt_input_lengths(H, K)
        self.msms, _ = W.msm_inventory(H, K)
        self.alg_ntt_bytes, self.alg_msm_bytes = W.algorithmic_bytes(H, K)
        D = max(3 * H - 1, K - 1)
        # base offsets: shifted_powers(d) = powers_of_g[max_degree - d ..]
        self.offsets = [0] * len(self.msms)
        for i, (n, what) in enumerate(self.msms):
            if "shifted" in what:
                self.offsets[i] = D - ((H - 2) if "g_1" in what else (K - 2))
        if bases is None:
            tau = np.array([0x9c5d7e2b4a6f8091, 0x1f3a, 0, 0], dtype=np.uint64)
            bases = M.Bases.srs_powers(tau, D + 1)
            bases.precompute()
        self.bases = bases
        self.srs = type("Srs", (), {"powers_of_g": bases})()
        rng = np.random.default_rng(seed)
        self.data = rand_fr_np(rng, 1 << max(lg for lg, _, _ in self.ntts))      # a Vec<Fr> of the largest domain
        self.scal = rand_fr_np(rng, K)
        self.out = np.zeros(18, dtype=np.uint64)
        self.ntt_s = self.msm_s = 0.0
        # the 15 MSMs by call: round 1 (w, z_a, z_b, mask), round 2 (t, g_1, g_1 shifted, h_1), round 3 (g_2, g_2 shifted, h_2),
        # opening at beta (witness, shifted witness), opening at gamma (witness, shifted witness); distinct host vectors per
        # polynomial, the SAME vector for a polynomial and its shifted commitment
        self.groups = [[0, 1, 2, 3], [4, 5, 6, 7], [8, 9, 10], [11, 12], [13, 14]]
        same_as = {6: 5, 9: 8}
        self.vecs = {}
        for i, (n, _) in enumerate(self.msms):
            self.vecs[i] = self.vecs[same_as[i]] if i in same_as else self.scal[(i * 4099) % 1024:][:n].copy()

    def step(self, dist=None, torch=None):
        t0 = time.perf_counter()
        for (lg, inverse, _), in_len in zip(self.ntts, self.in_lens):
            rc = self.lib.mh_ntt_len(self.curve, self.data.ctypes.data, in_len, lg, 1 if inverse else 0)
            assert rc == 0
        t1 = time.perf_counter()
        if self.batched:
            # one mh_msm_batch per PC::commit call (lib.rs:172,193,213) -- the labeled polynomials of a round -- and one per
            # opening point (lib.rs:292); a degree-bounded polynomial passes the same host vector twice
            import ctypes as C
            for grp in self.groups:
                k = len(grp)
                handles = (C.c_uint64 * k)(*[self.bases.handle] * k)
                offs = (C.c_size_t * k)(*[self.offsets[i] for i in grp])
                ptrs = (C.c_void_p * k)(*[self.vecs[i].ctypes.data for i in grp])
                ns = (C.c_size_t * k)(*[self.msms[i][0] for i in grp])
                out = np.zeros(18 * k, dtype=np.uint64)
                rc = self.lib.mh_msm_batch(k, handles, offs, ptrs, ns, 1, out.ctypes.data)
                assert rc == 0, self.lib.mh_last_error()
        else:
            for (n, _), off in zip(self.msms, self.offsets):
                rc = self.lib.mh_msm(self.bases.handle, off, self.scal.ctypes.data, 1, n, self.out.ctypes.data)
                assert rc == 0, self.lib.mh_last_error()
        t2 = time.perf_counter()
        self.ntt_s += t1 - t0
        self.msm_s += t2 - t1


def seam_route_measure(M, log_n, bases, steps=2, batched=True):
    sr = SeamRoute(M, log_n, bases, batched=batched)
    sr.step()                                        # warm-up: staging buffers, MSM workspace
    sr.ntt_s = sr.msm_s = 0.0
    for _ in range(steps):
        sr.step()
    H = 1 << log_n
    ntt_bytes = sum((32 << lg) + 32 * l for (lg, _, _), l in zip(sr.ntts, sr.in_lens))   # the caller's elements up, the whole domain down
    msm_bytes = sum(32 * n for n, _ in sr.msms)                        # scalars only: the bases are resident
    return {"ms_per_proof": round((sr.ntt_s + sr.msm_s) * 1e3 / steps, 2), "ntt_ms": round(sr.ntt_s * 1e3 / steps, 2),
            "msm_ms": round(sr.msm_s * 1e3 / steps, 2), "steps": steps,
            "msm_calls": "one mh_msm_batch per PC::commit / opening point (5 calls)" if batched else "15 separate mh_msm calls",
            "pcie_bytes_per_proof": ntt_bytes + msm_bytes,
            "pcie_floor_ms": round((ntt_bytes + msm_bytes) / 57e9 * 1e3, 1),
            "pcie_floor_note": "every byte of a transform crosses PCIe before (in) or after (out) its kernels and the calls are synchronous, "
                               "so bytes / ~57 GB/s (one direction at a time) bounds the route from below, whatever the kernels cost",
            "what": "seam route lower bound: the reference's 30 transforms through mh_ntt_len and 15 MSMs through mh_msm_batch / mh_msm with "
                    "HOST pointers (pageable, like Vec<Fr>), SRS + window table resident; the Rust host work between the "
                    "calls is not included.  Compare with ms_per_step of the device-resident prover (mh_marlin_prove_dev)",
            "constraints_per_s": round(H / ((sr.ntt_s + sr.msm_s) / steps), 1)}


def _cpu_inventory(cref, W, log_n, ntt_threads, msm_threads, rng):
    """NTT + MSM inventory of one prove at 2^log_n constraints on the C restatement:
    (seconds NTT, seconds MSM, busy threads of the MSM part)."""
    H = 1 << log_n
    K = 4 * H
    bases, _ = cref.bases_arith(K)
    ntts = W.ntt_inventory(H, K)
    msms, _ = W.msm_inventory(H, K)
    max_lg = max(lg for lg, _, _ in ntts)
    data = rand_fr_np(rng, 1 << max_lg)
    scal = rand_fr_np(rng, K)
    t0 = time.time()
    for lg, inverse, _ in ntts:
        cref.ntt(data[: 1 << lg], inverse=inverse, threads=ntt_threads)
    t_ntt = time.time() - t0
    t0 = time.time()
    for n, _ in msms:
        cref.msm(bases[:n], scal[:n], montgomery=True, threads=msm_threads)
    t_msm = time.time() - t0
    # the restatement parallelises an MSM like arkworks does -- one task per c-bit window -- so at most
    # ceil(255 / c) threads are ever busy (c = ceil(log2 n) * 69 / 100 + 2), whatever the host offers
    lg = (K - 1).bit_length()
    return t_ntt, t_msm, min(msm_threads, -(-255 // (lg * 69 // 100 + 2)))


def _usable_cpus():
    """Hardware threads this process may actually run on: the affinity mask, capped by a cgroup CPU quota if one is set
    (os.cpu_count() reports the machine, not the container)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def _best_ntt_threads(cref, usable, rng, log_n=18):
    """Fork-join over a stage's butterflies stops scaling long before a 256-thread host is full (a barrier per stage,
    2^17 butterflies per stage at this size): time one transform per candidate and keep the fastest."""
    data = rand_fr_np(rng, 1 << log_n)
    best = (None, 1)
    for t in sorted({min(usable, c) for c in (1, 4, 8, 16, 32, 64, 128, usable)}):
        cref.ntt(data, inverse=False, threads=t)                       # warm the thread team
        t0 = time.time()
        cref.ntt(data, inverse=False, threads=t)
        dt = time.time() - t0
        if best[0] is None or dt < best[0]:
            best = (dt, t)
    return best[1]


def cpu_baseline(log_n_all=18, log_n_one=13):
    """The C restatement (oracle/c/ref_hotpath.c, kind "port": arkworks' algorithm class -- radix-2 in-place NTT with the
    butterflies of a stage split over threads like ark-poly's rayon chunks, Pippenger with ark-ec's window rule and one
    task per window -- NOT arkworks itself) timed on this host: the NTT + MSM inventory of one prove (SURVEY.md Appendix A:
    30 transforms, 15 MSMs) at 2^log_n_all constraints (BASELINE configs[1]'s size; the headline 2^20 would take ~4x
    longer than the few minutes this run may use) with the thread counts that are fastest here, and with ONE thread at
    2^log_n_one."""
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")               # no spinning teams on an oversubscribed host
    from oracle import cref
    from marlin_amd import workload as W
    cref.build()
    host_threads = os.cpu_count() or 1
    usable = _usable_cpus()
    rng = np.random.default_rng(7)
    ntt_threads = _best_ntt_threads(cref, usable, rng)
    t_ntt1, t_msm1, _ = _cpu_inventory(cref, W, log_n_one, 1, 1, rng)
    t_ntt, t_msm, busy = _cpu_inventory(cref, W, log_n_all, ntt_threads, usable, rng)
    return {
        "value": (1 << log_n_all) / (t_ntt + t_msm), "unit": "constraints/s", "cores": max(ntt_threads, busy), "kind": "port",
        "cores_note": "NTT half on %d threads; MSM half: one task per window like ark-ec, so at most %d threads are ever busy "
                      "(%d usable on this host)" % (ntt_threads, busy, usable),
        "log_constraints": log_n_all,
        "host_threads": host_threads, "usable_threads": usable, "ntt_threads": ntt_threads, "msm_threads_busy": busy,
        "single_thread": {"value": (1 << log_n_one) / (t_ntt1 + t_msm1), "unit": "constraints/s", "cores": 1,
                          "sample": "same inventory at 2^%d constraints, 1 thread: NTT %.2fs, MSM %.2fs" % (log_n_one, t_ntt1, t_msm1)},
        "sample": "oracle/c/ref_hotpath.c (C restatement of arkworks' radix-2 NTT + Pippenger, NOT arkworks itself): "
                  "NTT+MSM inventory (30 transforms, 15 MSMs) of one prove at 2^%d constraints on a host with %d hardware "
                  "threads (%d usable): NTT %.2fs on %d threads (butterflies of each stage split over the team; the thread "
                  "count is the fastest of {1,4,...,all} measured on one 2^18 transform), MSM %.2fs (one task per window: %d "
                  "threads busy); witness synthesis, AHP glue and Fiat-Shamir not included on either side of the comparison"
                  % (log_n_all, host_threads, usable, t_ntt, ntt_threads, t_msm, busy),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--log-constraints", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", choices=["marlin-prove", "hotpath-inventory", "seam-route"], default=None)
    ap.add_argument("--full-prof", action="store_true",
                    help="record HIP events around every kernel family inside the timed region (costs ~1 ms per proof); default: "
                         "only the dominant kernel is timed there and the breakdown comes from extra untimed proofs")
    ap.add_argument("--no-verify", action="store_true", help="skip Marlin::verify of the last proof (host pairing, ~0.3 s, untimed)")
    ap.add_argument("--no-sliced", action="store_true",
                    help="multi-GPU: keep rounds 2 and 3 replicated on every rank (no distributed transforms / all-to-all)")
    ap.add_argument("--no-seam-route", action="store_true", help="skip the (untimed, ~1 s) seam-route measurement of the default run")
    ap.add_argument("--cpu-baseline-log", type=int, default=None,
                    help="log2 constraints of the CPU baseline's all-core sample (default: the size that is proved, at most 20 -- the "
                         "headline size takes about a minute on the GPU box's 16 usable threads; 18 takes ~20 s)")
    ap.add_argument("--transport", choices=["auto", "native", "callback"], default=os.environ.get("BENCH_TRANSPORT", "auto"),
                    help="multi-GPU exchanges: native = RCCL called by the library on its own stream (mh_marlin_set_rccl); callback = "
                         "torch.distributed through the registered Python callbacks; auto = native over RCCL, callback over gloo, and the "
                         "callback transport whenever the native one fails its self-test on any rank")
    ap.add_argument("--simulate-rank", default=None, metavar="R/G",
                    help="MEASUREMENT AID, one GPU: run what rank R of G would run (its bucket range of every MSM + the replicated "
                         "AHP rounds) with the exchange replaced by a local copy; the proof is not valid and the JSON line says so")
    ap.add_argument("--throughput", action="store_true",
                    help="also measure `throughput_pipelined` (untimed, ~20 s and a second key + window table in device memory): two independent "
                         "provers sharing the GPU.  Opt-in since round 6 (ADVICE r05): the default run no longer starts extra prover processes")
    ap.add_argument("--no-throughput", action="store_true", help="(accepted for older scripts; the measurement is off unless --throughput is given)")
    ap.add_argument("--rehearsal", action="store_true",
                    help="multi-GPU: accept that several ranks share a physical device (tests / tools/rehearse_ranks.sh on a one-GPU box); "
                         "without it a line whose ranks_seen hold fewer distinct devices than ranks carries value = null -- N ranks on "
                         "one GPU are not an N-GPU measurement")
    ap.add_argument("--pc", choices=["marlin", "sonic"], default="marlin",
                    help="polynomial commitment scheme (MarlinKZG10 = headline config; SonicKZG10 = configs[4]); "
                         "the curve is chosen with MARLIN_AMD_CURVE=bls12_381|bn254")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # started plainly with --gpus N: become N ranks (one per GPU) under torch.distributed.run on this node
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # the host driver only supports dmabuf IPC (RCCL needs it)
        sys.exit(subprocess.call(cmd, env=env))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # test hooks: BENCH_BACKEND=gloo and BENCH_SINGLE_DEVICE=1 let the N > 1 code path run with every rank on GPU 0
    # (the GPU box of this project has one GPU; RCCL refuses two ranks on one device)
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if os.environ.get("BENCH_SINGLE_DEVICE"):
        local_rank = 0
    import torch
    dist = None
    dry = bool(os.environ.get("BENCH_DRY_RUN"))     # CPU test hook: rank plumbing only, no device, no number
    if world != args.gpus and rank == 0:
        print("bench.py: --gpus %d but WORLD_SIZE=%d; the launcher's world size is what runs" % (args.gpus, world), file=sys.stderr)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if not dry:
            torch.cuda.set_device(local_rank)
        if backend == "nccl" and not dry:
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    if dry:
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        if world > 1:
            dist.barrier()
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        if rank == 0:
            print(json.dumps({"metric": "marlin_prove_constraints_per_sec", "value": None, "unit": "constraints/s", "n_gpus": world,
                              "steps": args.steps, "warmup": args.warmup, "dry_run": True, "max_over_ranks": float(t.item())}))
        if world > 1:
            dist.destroy_process_group()
        return
    import marlin_amd as M
    M.init(local_rank)
    torch.cuda.set_device(local_rank)

    class watchdog:
        """A collective that one rank never enters blocks its peers for good: bound the set-up phase, so that a wedged
        transport ends the job (non-zero exit: torch.distributed.run then stops the other ranks) instead of hanging it."""
        def __init__(self, seconds, what):
            import threading
            self.t = threading.Timer(seconds, self.fire)
            self.t.daemon = True
            self.what = what
        def fire(self):
            print("bench.py: rank %d: %s did not finish in time; aborting the job" % (rank, self.what), file=sys.stderr, flush=True)
            os._exit(3)
        def __enter__(self):
            self.t.start()
            return self
        def __exit__(self, *a):
            self.t.cancel()
            return False

    workload = args.workload or "marlin-prove"
    if workload == "marlin-prove":
        wl = MarlinProve(M, args.log_constraints, args.pc)
        transport = None
        if world > 1:
            from marlin_amd import dist as MD

            def all_ranks(ok):
                flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cuda" if backend == "nccl" else "cpu")
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                return bool(int(flag.item()))
            want_sliced = not args.no_sliced and (world & (world - 1)) == 0 and world >= 4     # 2 ranks: the exchanges cost more than they save
            wd = watchdog(float(os.environ.get("BENCH_SETUP_TIMEOUT_S", "300")), "the set-up of the multi-GPU transport")
            wd.__enter__()
            if args.transport == "native" or (args.transport == "auto" and backend == "nccl"):
                # RCCL from C++ on the library's stream.  Every step that can fail is agreed on by all ranks before the next
                # collective is entered; any failure falls back to the callback transport below.
                try:
                    ok = MD.enable_native_rccl(dist, sliced=want_sliced)
                except Exception as e:
                    print("bench.py: native RCCL transport refused on rank %d: %s" % (rank, e), file=sys.stderr)
                    ok = False
                if all_ranks(ok):
                    try:
                        ok = MD.selftest_allgather(dist)
                    except Exception as e:
                        print("bench.py: native all-gather self-test raised on rank %d: %s" % (rank, e), file=sys.stderr)
                        ok = False
                    if all_ranks(ok):
                        transport = "native-rccl"
                        sliced_rounds = False
                        if want_sliced:
                            try:
                                ok = MD.selftest_alltoall(dist)
                            except Exception as e:
                                print("bench.py: native all-to-all self-test raised on rank %d: %s" % (rank, e), file=sys.stderr)
                                ok = False
                            sliced_rounds = all_ranks(ok)
                            if not sliced_rounds:
                                from marlin_amd import _lib as _ML
                                _ML.check(_ML.load().mh_marlin_rccl_sliced(0), "mh_marlin_rccl_sliced")
                                if rank == 0:
                                    print("bench.py: native all-to-all self-test failed; rounds 2 and 3 stay replicated", file=sys.stderr)
                        args.no_sliced = not sliced_rounds
                if transport is None:
                    MD.disable_sharded_prove()
                    if rank == 0:
                        print("bench.py: native RCCL transport unavailable; using the torch.distributed callbacks", file=sys.stderr)
        if world > 1 and transport is None:
            transport = "callback-torch.distributed-" + backend
            MD.enable_sharded_prove(dist, device=torch.device("cuda", local_rank) if backend == "nccl" else None)
            sliced_rounds = False
            if want_sliced:
                if backend == "nccl":
                    # the library runs on a stream torch knows, so that the all-to-alls are ordered with its kernels on the
                    # device instead of through two host synchronisations each
                    dev = torch.device("cuda", local_rank)
                    MD.enable_alltoall(dist, device=dev, stream=MD.use_torch_stream(dev))
                else:
                    MD.enable_alltoall(dist, device=None)
                # the sliced rounds depend on the all-to-all: check one distributed transform against the local one on every
                # rank, and fall back to the replicated rounds everywhere unless all of them agree
                try:
                    ok = MD.selftest_alltoall(dist)
                except Exception as e:
                    print("bench.py: all-to-all self-test raised on rank %d: %s" % (rank, e), file=sys.stderr)
                    ok = False
                flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cuda" if backend == "nccl" else "cpu")
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                sliced_rounds = bool(int(flag.item()))
                if not sliced_rounds:
                    from marlin_amd import _lib as _ML
                    _ML.check(_ML.load().mh_marlin_set_alltoall(None, None), "mh_marlin_set_alltoall")
                    if rank == 0:
                        print("bench.py: all-to-all self-test failed; rounds 2 and 3 stay replicated", file=sys.stderr)
            args.no_sliced = not sliced_rounds
        if world > 1:
            wd.__exit__()
        if world > 1:
            pass
        elif args.simulate_rank:
            from marlin_amd import dist as MD
            sr, sg = (int(x) for x in args.simulate_rank.split("/"))
            if sg < 4:
                args.no_sliced = True               # the library keeps the rounds replicated below 4 ranks
            if args.transport == "native":
                # the native transport's own code path without peers: the stand-in for librccl in its solo mode turns every
                # collective into a stream-ordered local copy issued from C++ (no Python in the exchange, unlike the callbacks below)
                mock = os.path.join(ROOT, "tests", "mock_rccl", "libmock_rccl.so")
                if not os.path.exists(mock):
                    sys.exit("bench.py: --simulate-rank with --transport native needs tests/mock_rccl/libmock_rccl.so (build())")
                os.environ["MH_RCCL_LIB"] = mock
                os.environ["MH_MOCK_RCCL_SOLO"] = "1"
                MD.enable_native_rccl_solo(sr, sg, sliced=not args.no_sliced)
            elif args.no_sliced:
                MD.enable_simulated_shard(sr, sg)
            else:
                MD.enable_simulated_alltoall(sr, sg, stream_ordered=not os.environ.get("BENCH_SIM_SYNC_EXCHANGE"))
    elif workload == "seam-route":
        wl = SeamRoute(M, args.log_constraints)
    else:
        wl = HotPathInventory(M, args.log_constraints, rank, world)

    def barrier():
        M.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        wl.step(dist, torch)
    # Timed region: HIP events around the dominant kernel only (family 2, the bucket accumulation: what `roofline` needs).
    # An event pair around each of the ~90 kernel-family scopes of a proof costs ~1 ms of launch gaps per proof
    # (profiles/r03y_*: 78.4 -> 77.5 ms on one GPU, 20.1 -> 19.1 ms on a rank of 8), so the per-family breakdown is taken from
    # `breakdown_steps` further, untimed proofs below.
    M.prof_enable(True, families=[2] if not args.full_prof else None)
    M.prof_reset()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        wl.step(dist, torch)
    barrier()
    elapsed = time.perf_counter() - t0
    acc_ms, acc_launches = M.prof_get(2)
    local_elapsed = elapsed
    breakdown_steps = args.steps
    if not args.full_prof:
        breakdown_steps = max(1, min(3, args.steps))
        M.prof_enable(True)
        M.prof_reset()
        if world > 1 or args.simulate_rank:
            from marlin_amd import dist as _MDx
            _MDx.exchange_stats(reset=True)
        barrier()
        tb0 = time.perf_counter()
        for _ in range(breakdown_steps):
            wl.step(dist, torch)
        barrier()
        breakdown_ms_per_step = (time.perf_counter() - tb0) * 1e3 / breakdown_steps
        acc_b_ms, _ = M.prof_get(2)
    else:
        breakdown_ms_per_step = elapsed * 1e3 / args.steps
        acc_b_ms = acc_ms
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # the same proof through the HOST-pointer entry point (mh_marlin_prove: instance + witness cross PCIe inside the call, 32 B per
    # constraint) -- what a caller whose witness lives in a Vec<Fr> pays; reported beside the headline, never as `value`
    # (benches/bench.rs:94-107 times prove with the circuit in host memory; its witness synthesis is outside BOTH numbers here)
    host_inputs_ms = None
    if workload == "marlin-prove" and world == 1 and not args.simulate_rank and not wl.host_inputs:
        M.prof_enable(False)
        hsteps = max(1, min(3, args.steps))
        wl.host_inputs = True
        wl.step(dist, torch)
        barrier()
        th0 = time.perf_counter()
        for _ in range(hsteps):
            wl.step(dist, torch)
        barrier()
        host_inputs_ms = (time.perf_counter() - th0) * 1e3 / hsteps
        wl.host_inputs = False

    # which physical devices took part: every rank reports (rank, local device, PCI bus id / uuid), all-gathered -- lets a
    # scaling record prove that N distinct GPUs ran
    def _dev_ident():
        try:
            pr = torch.cuda.get_device_properties(local_rank)
            bus = "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", 0), getattr(pr, "pci_device_id", 0))
            return {"rank": rank, "local_device": local_rank, "pci": bus, "uuid": str(getattr(pr, "uuid", "")), "name": pr.name}
        except Exception as e:
            return {"rank": rank, "local_device": local_rank, "error": str(e)[:80]}
    ranks_seen = [_dev_ident()]
    if world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, ranks_seen[0])
        ranks_seen = gathered

    ms_per_step = elapsed * 1e3 / args.steps
    value = wl.N / (elapsed / args.steps)

    # ---- what was produced (outside the timed region): the last proof's fingerprint on every rank, and Marlin::verify of it
    # by the product's own host verifier (mh_marlin_verify: transcript replay + pairing check, no tau) on rank 0.  Inputs and
    # seeds are fixed, so the fingerprint of an N-GPU run must equal the 1-GPU one.
    proof_info = None
    if workload == "marlin-prove" and getattr(wl, "proof", None) is not None and not args.simulate_rank:
        import hashlib
        pb = bytes(wl.proof)
        proof_info = {"bytes": len(pb), "sha256_32": hashlib.sha256(pb).hexdigest()[:32]}
        # what the parity claim of this line rests on: a fixture arkworks itself wrote (shim/tests/parity.rs ->
        # tests/golden/arkworks_*.json, consumed by tests/test_arkworks_golden.py) or, while no Rust toolchain has produced one,
        # the CPU oracle's golden proofs only
        import glob
        ark = sorted(os.path.basename(f) for f in glob.glob(os.path.join(ROOT, "tests", "golden", "arkworks_*.json")))
        proof_info["arkworks_pin"] = ({"status": "present", "files": ark} if ark else
                                      {"status": "absent", "reason": "no tests/golden/arkworks_*.json: no Rust toolchain in this image has run "
                                       "shim/tests/parity.rs; byte-identity below is against the CPU oracle (oracle/), not against arkworks"})
        if world > 1:
            hs = [None] * world
            dist.all_gather_object(hs, proof_info["sha256_32"])
            proof_info["identical_on_all_ranks"] = len(set(hs)) == 1
        if rank == 0:
            # a committed fixture, not oracle code: the CPU oracle's whole proof for these very inputs (BLS12-381 + MarlinKZG10)
            try:
                from marlin_amd import _lib as _L0
                gname = ("marlin_proofs_xl.json" if (_L0.CURVE, args.pc) == ("bls12_381", "marlin")
                         else "marlin_proofs_xl_%s_%s.json" % (_L0.CURVE, args.pc))
                gpath = os.path.join(ROOT, "tests", "golden", gname)
                gold = json.load(open(gpath)) if os.path.exists(gpath) else {"cases": []}
                same_inputs = bool(gold["cases"]) and (int(gold["tau"], 16), int(gold["gamma"], 16), bytes.fromhex(gold["zk_seed"])) == (wl.tau, wl.gamma, wl.seed) \
                    and (gold.get("curve", "bls12_381"), gold.get("pc", "marlin")) == (_L0.CURVE, args.pc)
                for case in gold["cases"]:
                    if (same_inputs and case["num_constraints"] == wl.N
                            and case["num_variables"] == 10 and (int(case["a"], 16), int(case["b"], 16)) == (wl.a, wl.b)):
                        proof_info["oracle_golden"] = {"file": "tests/golden/" + gname, "producer": gold.get("producer"),
                                                       "byte_identical": pb.hex() == case["proof_bytes"]}
            except Exception as e:
                proof_info["oracle_golden"] = {"error": str(e)[:200]}
        if rank == 0 and not args.no_verify:
            try:
                els = wl.srs.verifier_key(wl.pk, wl.GM.g2_generator_mont(), pc=args.pc)
                t0 = time.time()
                proof_info["verified"] = bool(wl.GM.verify(wl.pk.vk_bytes(), *els, wl.inst[1:], pb, pc=args.pc))
                proof_info["verify_host_s"] = round(time.time() - t0, 3)
            except Exception as e:                  # a check after the measurement must not cost the line
                proof_info["verify_error"] = str(e)[:200]

    # ---- roofline of the dominant kernel (MSM bucket accumulation), live HIP-event timing ----
    ntt_ms, ntt_launches = M.prof_get(0)
    ntt_side_ms, _ = M.prof_get(6)      # the transforms that ran beside round 1's bucket reduction (their own family: concurrent with msm)
    msm_ms, _ = M.prof_get(1)
    glue_ms, _ = M.prof_get(3)
    stages_ms, _ = M.prof_get(4)      # sort + bucket reduction by themselves
    reduce_ms, reduce_launches = M.prof_get(7)   # the bucket reduction by itself (rsum + plane kernels), part of family 4
    # pairs the timed launches really process: the prover folds each opening's shifted witness into the witness MSM
    from marlin_amd import workload as W
    msms_run = W.msm_executed(wl.N, pc=args.pc) if workload == "marlin-prove" else wl.msms
    # bucket-range sharding: every rank handles all pairs' digits that fall into its 1/world of the buckets (a simulated rank of
    # G counts as one of G)
    world_eff = int(args.simulate_rank.split("/")[1]) if args.simulate_rank else world
    msm_pairs_rank = sum(abs(n) for n, _ in msms_run) / world_eff
    # algorithmic bytes per (scalar, base) pair: a 32-byte scalar and an affine point of two Fq coordinates --
    # 128 B on BLS12-381 (Fq = 48 B), 96 B on BN254 (Fq = 32 B)
    from marlin_amd import _lib as _Lc
    pair_bytes = 32.0 + 2 * 8 * _Lc.FQ_LIMBS
    # the MSMs of a commit round run as one batched launch: bytes per launch = all pairs of the step / launches of the step
    bytes_per_launch = pair_bytes * msm_pairs_rank * args.steps / max(1, acc_launches)
    avg_launch_ms = acc_ms / max(1, acc_launches)
    achieved = bytes_per_launch / (avg_launch_ms * 1e-3) / 1e9 if avg_launch_ms > 0 else 0.0
    # HBM traffic of the accumulate kernel comes from a SEPARATE rocprofv3 --pmc capture (tools/profile.sh: counters cannot
    # be read inside this process), i.e. from another run -- usually another box -- than the one timed here; traffic_source
    # says which.  The capture is scaled to this run's launch size (bytes per input pair x pairs per launch).
    traffic, traffic_source = None, None
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc):
        try:
            pj = json.load(open(pmc))
            per_pair = pj.get("msm_accum_bytes_per_pair")
            if per_pair:
                traffic = per_pair * msm_pairs_rank * args.steps / max(1, acc_launches)
            else:
                traffic = pj.get("msm_accum_bytes_per_launch")
            import hashlib
            from marlin_amd import _lib as _Lh
            try:
                loaded = hashlib.sha256(open(_Lh.LIB_PATH, "rb").read()).hexdigest()[:16]
            except Exception:
                loaded = None
            cap_build = (pj.get("build") or {}).get("libmarlin_hip.so_sha256_16")
            traffic_source = {"file": "profiles/pmc_traffic.json", "capture": pj.get("source"), "box": pj.get("box"),
                              "build": pj.get("build"), "fetch_size_factor": pj.get("fetch_size_factor"),
                              "loaded_library_sha256_16": loaded,
                              "capture_is_of_the_loaded_build": bool(loaded and cap_build and loaded == cap_build),
                              "same_run_as_timing": False,
                              "note": "counters need their own rocprofv3 --pmc passes (a separate run of this very bench command, tools/profile.sh); "
                                      "capture_is_of_the_loaded_build says whether that run used the library that is loaded now"}
        except Exception:
            traffic, traffic_source = None, None
    # NTT bytes: the transforms this workload executes (the prover runs 16 of the reference's 30, see workload.ntt_executed)
    if workload == "marlin-prove":
        from marlin_amd import workload as _W
        ntt_bytes, ntt_what = _W.executed_ntt_bytes(wl.N), "16 executed transforms (64 B per point; the reference's 30 would be %.2f GB)" % (wl.alg_ntt_bytes / 1e9)
    else:
        ntt_bytes, ntt_what = wl.alg_ntt_bytes, "30 transforms of the inventory (64 B per point)"
    # secondary view: the kernel's real bound is VALU issue.  The window plan issues W bucket additions per input pair
    # (W = 13 with the fixed-base table at c = 20, 16 on the variable-base path at c = 16).
    tab_c, tab_w = (0, 0)
    if hasattr(wl, "srs"):
        tab_c, tab_w, _ = wl.srs.powers_of_g.table_info()
    path = "fixed-base" if tab_w else "variable-base"
    W_windows = tab_w or 16
    madds_per_s = (msm_pairs_rank * W_windows * args.steps) / (acc_ms * 1e-3) if acc_ms > 0 else 0.0
    mixes = accum_mix(_Lc.CURVE)
    mix = (mixes or {}).get(path)
    if mix is None:         # no ISA count for this curve / path: report the achieved rate without a bound rather than a wrong fraction
        mix, bound_adds = {"mad_u64": 0, "half_rate_other": 0, "full_rate": 0}, float("nan")
    else:
        bound_adds = valu_bound_adds_per_s(mix)
    valu = {"bound": "valu-issue", "kernel": "msmfb::accum30v_kernel" if tab_w else "msm::accum_kernel",
            "achieved": round(madds_per_s / 1e9, 3), "peak": round(bound_adds / 1e9, 3) if bound_adds == bound_adds else None, "unit": "G bucket additions/s",
            "frac": round(madds_per_s / bound_adds, 4) if bound_adds == bound_adds else None, "windows": W_windows, "window_bits": tab_c or 16,
            "instr_mix_source": "profiles/accum_isa_mix.json [%s]" % _Lc.CURVE if mixes else "built-in constants",
            "valu_instr_per_add": sum(mix.values()), "v_mad_u64_u32_per_add": mix["mad_u64"], "instr_mix_per_add": mix,
            "class_rate_T_lane_ops_per_s": VALU_CLASS_RATE_T,
            "note": "peak = 1 / sum(instructions of one bucket addition in a class / measured issue rate of the class): "
                    "v_mad_u64_u32 and the other multiply / carry / 64-bit instructions issue at half the rate of plain "
                    "32-bit VALU ops on gfx950 (profiles/r02a_microbench_controls.txt); mix counted in the ISA "
                    "(profiles/r02b_accum_loop_isa.txt)"}
    roofline = {"bound": "hbm", "kernel": valu["kernel"], "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_source,
                "algorithmic_bytes_per_launch": bytes_per_launch, "avg_launch_ms": round(avg_launch_ms, 4),
                "algorithmic_bytes_per_pair": pair_bytes,
                "note": "algorithmic bytes = %d B per (scalar, base) pair (SURVEY.md 8d); the MSM is VALU-issue bound "
                        "(%d VALU instr per bucket addition, %d additions per pair), see roofline_valu and DESIGN.md; "
                        "NTT family: %.1f GB/s over the %s" % (
                            int(pair_bytes), sum(mix.values()), W_windows,
                            (ntt_bytes * breakdown_steps) / ((ntt_ms + ntt_side_ms) * 1e-3) / 1e9 if ntt_ms > 0 else 0.0, ntt_what)}

    # ---- the bucket reduction (msm_fb.cuh: rsum_kernel + plane_kernel): general additions against the VALU-issue bound of their
    # instruction mix.  Algorithmic work: 2 general additions per owned bucket (one into its row sum, one into its column sum);
    # the butterflies, the bit planes and the host's ~40 operations per job come on top and are what `frac` loses.
    roofline_reduce = None
    red_mix = (mixes or {}).get("reduce")
    if tab_w and reduce_ms > 0 and red_mix:
        buckets_step = len(msms_run) * (1 << (tab_c - 1)) / world_eff
        gadds_per_s = 2.0 * buckets_step * breakdown_steps / (reduce_ms * 1e-3)
        red_bound = valu_bound_adds_per_s(red_mix)
        roofline_reduce = {"bound": "valu-issue", "kernel": "msmfb::rsum_kernel + msmfb::plane_kernel", "achieved": round(gadds_per_s / 1e9, 3),
                           "peak": round(red_bound / 1e9, 3), "unit": "G general additions/s (XYZZ += XYZZ)", "frac": round(gadds_per_s / red_bound, 4),
                           "ms_per_step": round(reduce_ms / breakdown_steps, 3), "launch_pairs_per_step": reduce_launches / breakdown_steps,
                           "buckets_per_step": buckets_step, "additions_per_bucket": 2,
                           "waves_per_simd": "2 (220 registers; 1 where a lane would get fewer than 16 buckets)",
                           "valu_instr_per_add": sum(red_mix.values()), "instr_mix_per_add": red_mix,
                           "instr_mix_source": "profiles/accum_isa_mix.json [%s][reduce]" % _Lc.CURVE,
                           "note": "round 4's segment reduction (running sums + a ~19-bit double-and-add per thread + tree) took 6.83 ms per proof "
                                   "at 2^20 = 0.35 of this bound; row / column sums + bit planes need no device-side point multiplication"}

    # ---- every rank's own view (a SCALE record has to explain itself): its wall time per step, its kernel families, and what
    # the exchanges cost it -- events on the library's stream around each collective (family 5) and the host's wall clock inside them
    exch_ms, exch_n = M.prof_get(5)
    side_ms, _ = M.prof_get(6)          # transforms on the second stream beside round 1's bucket reduction: concurrent with `msm`
    exch_calls, exch_host_ms = (0, 0.0)
    if world > 1 or args.simulate_rank:
        from marlin_amd import dist as _MDx
        exch_calls, exch_host_ms = _MDx.exchange_stats()
    mine = {"rank": rank, "ms_per_step": round(local_elapsed * 1e3 / args.steps, 3),
            "breakdown_ms_per_step": {"ntt": round(ntt_ms / breakdown_steps, 3), "msm": round(msm_ms / breakdown_steps, 3),
                                      "msm_accum": round(acc_ms / args.steps, 3), "msm_sort_and_reduce_stages": round(stages_ms / breakdown_steps, 3),
                                      "glue": round(glue_ms / breakdown_steps, 3), "exchange_on_stream": round(exch_ms / breakdown_steps, 3),
                                      "exchange_host_wall": round(exch_host_ms / breakdown_steps, 3),
                                      "exchanges_per_step": round(exch_calls / breakdown_steps, 2),
                                      "step_with_all_events": round(breakdown_ms_per_step, 3)}}
    per_rank = [mine]
    if world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        per_rank = gathered
    transport_info = None
    if workload == "marlin-prove" and (world > 1 or args.simulate_rank):
        from marlin_amd import dist as _MDx
        ni = _MDx.native_rccl_info()
        transport_info = {"kind": transport if world > 1 else ("simulated: native transport over the solo stand-in (local copies issued from C++)"
                                                               if ni["active"] else "simulated: Python callbacks (local copies)"),
                          "native_rccl": ni if ni["active"] else None}

    out = {
        "metric": "marlin_prove_constraints_per_sec", "value": round(value, 1), "unit": "constraints/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u32-limb Montgomery integers (Fr 256-bit in 8 x 32 and 9 x 30 bits; Fq 384-bit in 12 x 32 and 13 x 30 bits)",
        "data": "synthetic",
        "config": {"workload": ("marlin-prove: Marlin::prove from the padded R1CS instance + witness on (AHP rounds + KZG10 commit/open + "
                                "Fiat-Shamir; witness synthesis src/ahp/prover.rs:217-230 excluded), "
                                if workload == "marlin-prove" else
                                "hotpath-inventory: 30 NTT + 15 MSM of one Marlin::prove on synthetic vectors, ")
                               + "DummyCircuit 2^%d constraints, BLS12-381, MarlinKZG10 (benches/bench.rs shape; SURVEY.md Appendix A)"
                               % args.log_constraints,
                   "constraints": wl.N, "curve": "BLS12-381", "pc": "MarlinKZG10",
                   "parallelism": ("msm sharded by bucket range x%d (one all_gather of partial points per commit round), " % world) +
                                  ("AHP rounds replicated" if world == 1 or args.no_sliced else
                                   "rounds 2 and 3 on slices (distributed transforms with one all-to-all each, one all-gather of the round's polynomials), "
                                   "opening polynomials built, divided and multiplied on blocks of the SRS index space (MarlinKZG10), round 1 replicated")},
        "breakdown_ms_per_step": {"ntt": round(ntt_ms / breakdown_steps, 3), "msm": round(msm_ms / breakdown_steps, 3),
                                  "msm_accum": round(acc_ms / args.steps, 3),
                                  "msm_sort_and_reduce_stages": round(stages_ms / breakdown_steps, 3),
                                  "msm_hidden_under_accum": round(max(0.0, acc_b_ms + stages_ms - msm_ms) / breakdown_steps, 3),
                                  "glue": round(glue_ms / breakdown_steps, 3),
                                  "exchange": round(exch_ms / breakdown_steps, 3),
                                  "ntt_beside_msm_reduction": round(side_ms / breakdown_steps, 3),
                                  "host_and_other": round(breakdown_ms_per_step - (ntt_ms + msm_ms + glue_ms + exch_ms) / breakdown_steps, 3),
                                  "measured_on": ("the timed steps" if args.full_prof else
                                                  "%d further untimed proofs with events around every kernel family (%.3f ms each: the events cost "
                                                  "launch gaps, so the timed steps record the accumulate kernel only); msm_accum: the timed steps"
                                                  % (breakdown_steps, breakdown_ms_per_step))},
        "roofline": roofline,
        "roofline_valu": valu,
        "roofline_reduce": roofline_reduce,
        "host_inputs_ms_per_step": round(host_inputs_ms, 3) if host_inputs_ms is not None else None,
        "host_inputs_note": ("mh_marlin_prove with HOST pointers: the formatted input and the witness (32 B per constraint) cross PCIe inside the "
                             "call; ms_per_step / value above are mh_marlin_prove_dev, inputs already in HBM (the contract's timed region)"
                             if host_inputs_ms is not None else None),
        "accum_launches_per_step": acc_launches / max(1, args.steps),
        "proof": proof_info,
        "per_rank": per_rank if (world > 1 or args.simulate_rank) else None,
        "transport": transport_info,
        "ranks_seen": ranks_seen,
        "distinct_devices": len({(r.get("pci"), r.get("uuid")) for r in ranks_seen}),
    }
    from marlin_amd import _lib as _L
    curve_name = {"bls12_381": "BLS12-381", "bn254": "BN254"}[_L.CURVE]
    out["config"]["curve"] = curve_name
    out["config"]["pc"] = {"marlin": "MarlinKZG10", "sonic": "SonicKZG10"}[args.pc]
    out["config"]["workload"] = out["config"]["workload"].replace("BLS12-381, MarlinKZG10", "%s, %s" % (curve_name, out["config"]["pc"]))
    if _L.CURVE != "bls12_381" or args.pc != "marlin":
        out["dtype"] = "u32-limb Montgomery integers (Fr 256-bit, Fq %d-bit)" % (64 * _L.FQ_LIMBS)
        args.no_cpu_baseline = True          # the C restatement covers the headline configuration only
    if world > 1 and out["distinct_devices"] < world:
        out["rehearsal"] = bool(args.rehearsal)
        if not args.rehearsal:
            out["value"] = None
            out["note"] = ("%d ranks ran on %d distinct device(s): not an N-GPU measurement, so no scaling value is printed "
                           "(--rehearsal accepts shared devices for tests)" % (world, out["distinct_devices"]))
    if args.simulate_rank:
        out["simulated_rank"] = args.simulate_rank
        out["simulated_sliced_rounds"] = not args.no_sliced
        out["value"] = None
        out["note"] = ("simulation of one rank of a multi-GPU run on one GPU (exchange replaced by a local copy): ms_per_step is "
                       "that rank's time without the all_gather; the proofs made are not valid; not a benchmark result")
        args.no_cpu_baseline = True
    if workload == "seam-route":
        out["metric"] = "marlin_seam_route_constraints_per_sec"
        out["config"]["workload"] = ("seam-route: the reference's 30 NTTs through mh_ntt and 15 MSMs through mh_msm with HOST pointers "
                                     "(north_star's literal route; lower bound, Rust host work excluded), DummyCircuit 2^%d, %s" % (args.log_constraints, curve_name))
        out["breakdown_ms_per_step"]["host_pointer_ntt_calls"] = round(wl.ntt_s * 1e3 / (args.steps + args.warmup), 3)
        out["breakdown_ms_per_step"]["host_pointer_msm_calls"] = round(wl.msm_s * 1e3 / (args.steps + args.warmup), 3)
        args.no_cpu_baseline = True
    elif workload == "marlin-prove" and rank == 0 and world == 1 and not args.no_seam_route and not args.simulate_rank:
        try:
            out["seam_route"] = seam_route_measure(M, args.log_constraints, wl.srs.powers_of_g)
            out["seam_route"]["unbatched_msm_ms"] = seam_route_measure(M, args.log_constraints, wl.srs.powers_of_g, steps=1, batched=False)["msm_ms"]
        except Exception as e:                      # a side measurement must not cost the headline line
            out["seam_route"] = {"error": str(e)[:200]}
    if workload == "marlin-prove" and rank == 0 and world == 1 and args.throughput and not args.no_throughput and not args.simulate_rank:
        # NOT the headline: what the GPU delivers when independent provers (separate processes, each with its own key and window
        # table) share it -- one's host round trips and latency-bound stages are filled by the others' kernels (VERDICT r04 item 9,
        # in the form that needs no double-buffered prover).  benches/bench.rs runs its proofs one at a time, and so does `value`.
        try:
            import subprocess
            M.synchronize()
            tp = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "throughput_procs.py"), str(args.log_constraints), "2", "--pc", args.pc,
                                 "--secs", "3", "--json"], capture_output=True, text=True, timeout=300)
            if tp.returncode != 0:
                raise RuntimeError("tools/throughput_procs.py exited with %d: %s" % (tp.returncode, tp.stderr[-300:]))
            r2 = json.loads(tp.stdout.strip().splitlines()[-1])["2"]
            out["throughput_pipelined"] = dict(r2, vs_one_proof_at_a_time=round(r2["constraints_per_s"] / value, 4),
                                               what="two independent prover PROCESSES sharing this GPU (own context, key, window table), proving back to back for "
                                                    "3 s after warm-up: aggregate rate; not the headline (`value` = one proof at a time, as benches/bench.rs runs them)")
        except Exception as e:                      # a side measurement must not cost the headline line
            out["throughput_pipelined"] = {"error": str(e)[:200]}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # both sides of the line on the same workload: the size that was proved (the headline 2^20 takes about a minute here)
        out["cpu_baseline"] = cpu_baseline(log_n_all=args.cpu_baseline_log or min(args.log_constraints, 20))
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
