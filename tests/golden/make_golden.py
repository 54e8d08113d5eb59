#!/usr/bin/env python3
"""Generates tests/golden/marlin_proofs.json with the pure-Python oracle (oracle/marlin.py).

No arkworks output can be produced in this environment (no Rust toolchain), so
these are NOT arkworks-produced vectors: they freeze the oracle's restatement so that (a) oracle
drift is detected by the CPU tests and (b) the HIP prover is compared byte-for-byte against them
on the GPU box.  They are SELF-CONSISTENCY vectors (parity with arkworks bytes stays unpinned).
Run from the repo root:
    python tests/golden/make_golden.py            # the src/test.rs shapes (marlin_proofs.json, ~1 min)
    python tests/golden/make_golden.py large      # DummyCircuit at 2^10 (BASELINE configs[0]), 2^12, 2^14
                                                  # (marlin_proofs_large.json, ~15 min of pure-Python big-int work)
    python tests/golden/make_golden.py xl [logs]  # DummyCircuit at 2^16, 2^18 (BASELINE configs[1]) [, 2^20 = configs[2]] with
                                                  # the oracle's C backend for NTT / MSM / SRS (oracle/accel.py; the AHP
                                                  # rounds, the PC logic and Fiat-Shamir stay Python): marlin_proofs_xl.json.
                                                  # Before it writes anything it re-proves the 2^12 case of the pure-Python
                                                  # file with the backend on and checks the bytes.
    [ORACLE_CURVE=bn254] python tests/golden/make_golden.py xl-cfg <marlin|sonic> <logs>
                                                  # the same for another PC scheme / the other curve:
                                                  # marlin_proofs_xl_<curve>_<pc>.json (BASELINE configs[4], benches/bench.rs's shape)
"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ahp as AHP, marlin as MR, fs as FS, fields as F  # noqa: E402

TAU, GAMMA = 0x1f3a9c5d7e2b4a6f8091a2b3c4d5e6f708192a3b4c5d6e7f, 0x5eed5eed5eed5eed0123456789abcdef
ZK_SEED = bytes(range(32))

CASES = [
    # kind, num_constraints, num_variables   (src/test.rs:165-203 shapes + benches/bench.rs DummyCircuit)
    ("test_circuit", 25, 25), ("test_circuit", 26, 25), ("test_circuit", 25, 26),
    ("test_circuit", 100, 25), ("test_circuit", 25, 100),
    ("dummy_circuit", 32, 10), ("dummy_circuit", 64, 10),
]
LARGE_CASES = [("dummy_circuit", 1 << 10, 10), ("dummy_circuit", 1 << 12, 10), ("dummy_circuit", 1 << 14, 10)]


def build(kind, nc, nv):
    rng = FS.test_rng()
    a, b = FS.fr_rand(rng), FS.fr_rand(rng)
    if kind == "test_circuit":
        cs = AHP.finalize_test_circuit(AHP.test_circuit(a, b, nc, nv))
        pub = [a * b % F.R_MOD, a * b % F.R_MOD * b % F.R_MOD]
    else:
        cs = AHP.dummy_circuit(a, b, nv, nc)
        pub = [a * b % F.R_MOD]
    cs = AHP.pad_and_square(cs)
    return a, b, cs, pub


def main():
    large = len(sys.argv) > 1 and sys.argv[1] == "large"
    cases, fname = (LARGE_CASES, "marlin_proofs_large.json") if large else (CASES, "marlin_proofs.json")
    if len(sys.argv) > 1 and sys.argv[1] == "xl":
        from oracle import accel
        accel.enable()
        # the backend must reproduce the pure-Python oracle before its output is trusted as a fixture
        ref = json.load(open(os.path.join(ROOT, "tests", "golden", "marlin_proofs_large.json")))
        c12 = [c for c in ref["cases"] if c["num_constraints"] == 1 << 12][0]
        a, b, cs, pub = build("dummy_circuit", 1 << 12, 10)
        srs = MR.universal_setup(1 << 12, 1 << 12, 3 << 12, TAU, GAMMA)
        pk = MR.marlin_index(srs, cs)
        assert MR.proof_bytes(MR.prove(pk, cs, FS.ChaChaRng(ZK_SEED, 20))).hex() == c12["proof_bytes"], "C backend diverges from the pure-Python oracle"
        logs = [int(x) for x in sys.argv[2:]] or [16, 18]
        cases, fname = [("dummy_circuit", 1 << lg, 10) for lg in logs], "marlin_proofs_xl.json"
    pc = "marlin"
    if len(sys.argv) > 1 and sys.argv[1] == "xl-cfg":
        # another PC scheme and / or (ORACLE_CURVE=bn254) the other curve: BASELINE configs[4] and benches/bench.rs's own shape.
        # No pure-Python fixture of these configurations exists at 2^12, so the backend is first checked against the
        # pure-Python oracle on a 2^9 proof of the same configuration.
        from oracle import accel
        pc = sys.argv[2]
        a, b, cs, pub = build("dummy_circuit", 1 << 9, 10)
        srs = MR.universal_setup(1 << 9, 1 << 9, 3 << 9, TAU, GAMMA)
        want = MR.proof_bytes(MR.prove(MR.marlin_index(srs, cs, pc=pc), cs, FS.ChaChaRng(ZK_SEED, 20)))
        accel.enable()
        srs = MR.universal_setup(1 << 9, 1 << 9, 3 << 9, TAU, GAMMA)
        got = MR.proof_bytes(MR.prove(MR.marlin_index(srs, cs, pc=pc), cs, FS.ChaChaRng(ZK_SEED, 20)))
        assert got == want, "C backend diverges from the pure-Python oracle"
        logs = [int(x) for x in sys.argv[3:]]
        cases, fname = [("dummy_circuit", 1 << lg, 10) for lg in logs], "marlin_proofs_xl_%s_%s.json" % (F.CURVE, pc)
    out = {"tau": hex(TAU), "gamma": hex(GAMMA), "zk_seed": ZK_SEED.hex(), "zk_rng": "ChaCha20 (rand_chacha ChaChaRng::from_seed)",
           "cases": []}
    path = os.path.join(ROOT, "tests", "golden", fname)
    if fname.startswith("marlin_proofs_xl"):
        out["curve"], out["pc"] = F.CURVE, pc
        out["producer"] = "oracle/*.py with the C backend of oracle/accel.py for NTT / MSM / SRS powers (self-consistency vectors, not arkworks output)"
        if os.path.exists(path):                                  # add sizes to an existing file
            out = json.load(open(path))
            out["cases"] = [c for c in out["cases"] if (c["kind"], c["num_constraints"], c["num_variables"]) not in cases]
    for kind, nc, nv in cases:
        a, b, cs, pub = build(kind, nc, nv)
        nnz = 3 * max(nc, nv)
        srs = MR.universal_setup(max(nc, nv), max(nc, nv), nnz, TAU, GAMMA)
        pk = MR.marlin_index(srs, cs, pc=pc)
        pr = MR.prove(pk, cs, FS.ChaChaRng(ZK_SEED, 20))
        assert MR.verify(pk, pub, pr) and not MR.verify(pk, [a] * len(pub), pr)
        pb = MR.proof_bytes(pr)
        out["cases"].append({
            "kind": kind, "num_constraints": nc, "num_variables": nv, "a": hex(a), "b": hex(b),
            "H": pk.index.domain_h.size, "K": pk.index.domain_k.size, "srs_max_degree": srs.max_degree,
            "vk_bytes_blake2s": hashlib.blake2s(MR.vk_bytes(pk)).hexdigest(),
            "challenges": {k: hex(v) for k, v in pr.challenges.items()},
            "evaluations": [hex(e) for e in pr.evaluations],
            "proof_bytes": pb.hex(),
        })
        # blake2s of every prover polynomial's coefficient bytes: lets the GPU test say WHICH polynomial diverged
        out["cases"][-1]["poly_blake2s"] = {l: hashlib.blake2s(b"".join(FS.fr_bytes(x) for x in pr.polys[l])).hexdigest()
                                            for l in ("w", "z_a", "z_b", "mask_poly", "t", "g_1", "h_1", "g_2", "h_2")}
        print(kind, nc, nv, "H", pk.index.domain_h.size, "K", pk.index.domain_k.size, len(pb), "bytes", flush=True)
        json.dump(out, open(os.path.join(ROOT, "tests", "golden", fname), "w"), indent=1)


if __name__ == "__main__":
    main()
