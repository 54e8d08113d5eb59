#!/usr/bin/env python3
"""Writes tests/golden/oracle_in_arkworks_format.json: a fixture in EXACTLY the format shim/tests/parity.rs
(`write_self_contained_golden_vectors`) writes on a machine with cargo -- but produced by this repository's own oracle.

It pins nothing about arkworks (the producer field says so); it exists so that the consumer,
tests/test_arkworks_golden.py, is exercised end to end here -- field decoding, SRS through
`mh_bases_upload_serialized` or regenerated from tau, circuit construction, wire bytes -- and is known to work the day
an `arkworks_*.json` made by `cargo test` is dropped next to it.  Run from the repo root:
    python tests/golden/make_arkworks_format_fixture.py
"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ahp as AHP, marlin as MR, fs as FS, fields as F  # noqa: E402

TAU, GAMMA = 0x2b7e151628aed2a6abf7158809cf4f3c762e7160f38b4da56a784d9045190cfe % F.R_MOD, 0x3243f6a8885a308d313198a2e0370734
A, B = 0x1234567890abcdef1122334455667788, 0x0fedcba9876543210011223344556677
ZK_SEED = bytes([42] * 32)


def le32(x):
    return (x % F.R_MOD).to_bytes(32, "little").hex()


def case(name, pc, kind, nc, nv, setup, embed):
    if kind == "test":
        cs = AHP.finalize_test_circuit(AHP.test_circuit(A, B, nc, nv))
        pub = [A * B % F.R_MOD, A * B % F.R_MOD * B % F.R_MOD]
    else:
        cs = AHP.dummy_circuit(A, B, nv, nc)
        pub = [A * B % F.R_MOD]
    cs = AHP.pad_and_square(cs)
    srs = MR.universal_setup(setup[0], setup[1], setup[2], TAU, GAMMA)
    pk = MR.marlin_index(srs, cs, pc)
    pr = MR.prove(pk, cs, FS.ChaChaRng(ZK_SEED, 20))
    assert MR.verify(pk, pub, pr)
    g_bytes = b"".join(MR.g1_compressed(p) for p in srs.powers_of_g)
    gg_bytes = b"".join(MR.g1_compressed(srs.all_gamma[i]) for i in range(srs.max_degree + 2))
    s = {"num_constraints": setup[0], "num_variables": setup[1], "num_non_zero": setup[2], "max_degree": srs.max_degree,
         "tau": le32(TAU), "gamma": le32(GAMMA), "powers_of_g_blake2s": hashlib.blake2s(g_bytes).hexdigest(),
         "powers_of_gamma_g_blake2s": hashlib.blake2s(gg_bytes).hexdigest()}
    if embed:
        s["powers_of_g"], s["powers_of_gamma_g"] = g_bytes.hex(), gg_bytes.hex()
    return {"name": name, "curve": "bls12_381", "pc": pc,
            "circuit": {"kind": kind, "num_constraints": nc, "num_variables": nv, "a": le32(A), "b": le32(B)},
            "srs": s, "zk_seed": ZK_SEED.hex(), "zk_rounds": 20, "public_input": [le32(x) for x in pub],
            "vk_to_bytes": MR.vk_bytes(pk).hex(), "proof": MR.proof_serialize(pr).hex()}


def main():
    assert F.CURVE == "bls12_381"
    cases = []
    for name, nc, nv in [("tall_matrix_big", 100, 25), ("squat_matrix_small", 25, 26), ("square_matrix", 25, 25)]:
        for pc in ("marlin", "sonic"):
            cases.append(case(name, pc, "test", nc, nv, (100, max(nv, 25), 300), name == "tall_matrix_big"))
    cases.append(case("dummy_2p6", "marlin", "dummy", 64, 10, (64, 64, 192), False))
    out = {"format": 1, "producer": "ORACLE of this repository (tests/golden/make_arkworks_format_fixture.py) -- NOT arkworks; exercises the consumer only",
           "cases": cases}
    path = os.path.join(ROOT, "tests", "golden", "oracle_in_arkworks_format.json")
    json.dump(out, open(path, "w"), indent=0)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
