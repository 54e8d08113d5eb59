"""CPU tests of the protocol-level oracle (oracle/fs.py, ahp.py, marlin.py).

The reference holds no golden vectors; what its own tests assert are properties,
and those are what pins the restatement here: prove -> verify accepts and a wrong public input is
rejected (src/test.rs:158,161), both sumcheck LCs vanish (src/ahp/mod.rs:177,214 -- asserted inside
oracle.marlin.prove), degree bounds (prover.rs:385-388,516,556-557,697-698 -- asserted inside the
round functions), the Lagrange helper identities (src/ahp/mod.rs:340-387) and the arithmetisation
identities (constraint_systems.rs:389-404).  Public KATs pin ChaCha20 and Blake2s.
"""
import hashlib
import json
import os

from oracle import ahp as AHP, marlin as MR, fs as FS, fields as F, poly as OP

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = F.R_MOD


def test_chacha20_rfc7539_block_kat():
    key = list(range(32))
    import struct
    kw = list(struct.unpack("<8I", bytes(key)))
    # RFC 7539 §2.3.2: counter = 1, nonce = 00:00:00:09:00:00:00:4a:00:00:00:00
    out = FS.chacha_block(kw, 1 | (0x09000000 << 32), 0x4a000000, 20)
    assert out[0] == 0xe4e7f110 and out[1] == 0x15593bd1 and out[15] == 0x4e3c50a2


def test_blake2s_kat():
    # RFC 7693 Appendix B
    assert hashlib.blake2s(b"abc").hexdigest() == "508c5e8c327c14e2e1a72ba34eeb452f37458b209ed63a294d999b4c86675982"


def test_fr_rand_uses_raw_limbs_as_montgomery():
    rng = FS.ChaChaRng(bytes(32), 20)
    limbs = [rng.next_u64() for _ in range(4)]
    limbs[3] &= (1 << 63) - 1
    x = sum(l << (64 * i) for i, l in enumerate(limbs))
    v = FS.fr_rand(FS.ChaChaRng(bytes(32), 20))
    if x < R:
        assert v == x * F.FR_MONT_RINV % R


def test_bivariate_lagrange_helpers():
    """src/ahp/mod.rs:340-366."""
    for log in range(1, 7):
        d = OP.Domain(1 << log)
        els = d.elements()
        x = 0x123456789 + log
        fast = d.batch_eval_unnormalized_bivariate_lagrange_poly_with_diff_inputs(x)
        assert fast == [d.eval_unnormalized_bivariate_lagrange_poly(x, y) for y in els]
        same = d.batch_eval_unnormalized_bivariate_lagrange_poly_with_same_inputs()
        assert same == [d.eval_unnormalized_bivariate_lagrange_poly(y, y) for y in els]


def test_sumcheck_fact():
    """src/ahp/mod.rs:368-387: sum over H of p = |H| * (a_0 + a_n) for deg-16 p, |H| = 16."""
    d = OP.Domain(16)
    p = [(i * 7919 + 13) % R for i in range(17)]
    assert sum(OP.poly_eval(p, h) for h in d.elements()) % R == 16 * (p[0] + p[16]) % R


def test_arithmetization_identities():
    """constraint_systems.rs:389-404 on a small circuit."""
    rng = FS.test_rng()
    a, b = FS.fr_rand(rng), FS.fr_rand(rng)
    cs = AHP.pad_and_square(AHP.finalize_test_circuit(AHP.test_circuit(a, b, 6, 5)))
    idx = AHP.index(cs)
    dk, dh, dx = idx.domain_k, idx.domain_h, idx.domain_x
    for label, key in (("row", "row"), ("col", "col"), ("a_val", "val_a"), ("row_col", "row_col")):
        assert [OP.poly_eval(idx.polys[label], k) for k in dk.elements()] == idx.evals_on_K[key]
    els = dh.elements()
    inv = {e: i for i, e in enumerate(els)}
    eta = (3, 5, 7)
    for k in range(dk.size):
        col = inv[idx.evals_on_K["row"][k]]        # transposed
        row = inv[idx.evals_on_K["col"][k]]
        got = sum(e * idx.evals_on_K[v][k] for e, v in zip(eta, ("val_a", "val_b", "val_c"))) % R
        want = 0
        for e, m in zip(eta, (idx.a, idx.b, idx.c)):
            for f, j in m[row]:
                if dh.reindex_by_subdomain(dx, j) == col:
                    want += e * f
        u = dh.eval_unnormalized_bivariate_lagrange_poly(els[col], els[col])
        if k < idx.num_non_zero:
            assert got == want * pow(u, -1, R) % R


def _case(kind, nc, nv):
    rng = FS.test_rng()
    a, b = FS.fr_rand(rng), FS.fr_rand(rng)
    if kind == "test_circuit":
        cs = AHP.finalize_test_circuit(AHP.test_circuit(a, b, nc, nv))
        pub = [a * b % R, a * b % R * b % R]
    else:
        cs = AHP.dummy_circuit(a, b, nv, nc)
        pub = [a * b % R]
    return a, b, AHP.pad_and_square(cs), pub


def test_prove_verify_roundtrip_and_golden():
    """src/test.rs:132-163 on two of its shapes + DummyCircuit; also freezes the oracle against
    tests/golden/marlin_proofs.json (made by tests/golden/make_golden.py)."""
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "marlin_proofs.json")))
    tau, gamma, seed = int(g["tau"], 16), int(g["gamma"], 16), bytes.fromhex(g["zk_seed"])
    for case in g["cases"]:
        if (case["kind"], case["num_constraints"], case["num_variables"]) not in (
                ("test_circuit", 25, 25), ("test_circuit", 26, 25), ("dummy_circuit", 32, 10)):
            continue
        nc, nv = case["num_constraints"], case["num_variables"]
        a, b, cs, pub = _case(case["kind"], nc, nv)
        assert hex(a) == case["a"]
        srs = MR.universal_setup(max(nc, nv), max(nc, nv), 3 * max(nc, nv), tau, gamma)
        pk = MR.marlin_index(srs, cs)
        assert (pk.index.domain_h.size, pk.index.domain_k.size) == (case["H"], case["K"])
        pr = MR.prove(pk, cs, FS.ChaChaRng(seed, 20))
        assert MR.verify(pk, pub, pr)
        assert not MR.verify(pk, [a] * len(pub), pr)
        assert MR.proof_bytes(pr).hex() == case["proof_bytes"]
        # a tampered evaluation or commitment must be rejected
        pr.evaluations[0] = (pr.evaluations[0] + 1) % R
        assert not MR.verify(pk, pub, pr)


def test_reference_fixture_dimensions():
    """padded dimensions of the reference's five test shapes."""
    want = {(100, 25): (128, 512), (26, 25): (32, 128), (25, 100): (128, 128), (25, 26): (32, 128), (25, 25): (32, 128)}
    for (nc, nv), (H, K) in want.items():
        _, _, cs, _ = _case("test_circuit", nc, nv)
        idx_h = OP.Domain(cs.num_constraints).size
        nnz = sum(len(set([j for _, j in ra] + [j for _, j in rb] + [j for _, j in rc])) for ra, rb, rc in zip(cs.a, cs.b, cs.c))
        assert (idx_h, OP.Domain(nnz).size) == (H, K)


def test_c_backend_of_the_oracle_reproduces_the_pure_python_proof():
    """oracle/accel.py swaps the oracle's NTT (>= 2^10 points), its KZG10 MSM (>= 2^10 coefficients) and its SRS powers for
    the C restatement -- the backend behind tests/golden/marlin_proofs_xl.json (2^16 .. 2^20).  At 2^8 constraints (H = 2^8,
    K = 2^10: transforms of 2^10 and 2^11 points and K-sized MSMs take the C path) index and proof bytes must be the
    pure-Python oracle's, for both PC schemes."""
    from oracle import accel
    a, b, n = 0x1234567, 0x7654321, 1 << 8

    def run(pc):
        cs = AHP.pad_and_square(AHP.dummy_circuit(a, b, 10, n))
        srs = MR.universal_setup(n, n, 3 * n, 0x1f3a9c5d7e2b4a6f, 0x5eed5eed)
        pk = MR.marlin_index(srs, cs, pc)
        return MR.vk_bytes(pk), MR.proof_bytes(MR.prove(pk, cs, FS.ChaChaRng(bytes(range(32)), 20)))
    pure = {pc: run(pc) for pc in ("marlin", "sonic")}
    accel.enable(threads=2)
    try:
        fast = {pc: run(pc) for pc in ("marlin", "sonic")}
    finally:
        accel.disable()
    assert fast == pure
