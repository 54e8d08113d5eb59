"""Parity pins above the sizes the pure-Python oracle can prove at.

* byte-identical proofs against oracle-generated golden fixtures at 2^10 (BASELINE configs[0]), 2^12 and 2^14
  (tests/golden/marlin_proofs_large.json, tests/golden/make_golden.py large);
* the device's `DensePolynomial::rand` (rng.cuh: ChaCha blocks in parallel + rejection sampling as stream
  compaction) against the sequential `Fp256::rand` stream at 2^16 / 2^18 (3|H| draws, prover.rs:370-380);
* at 2^18 (configs[1]) and 2^20 (configs[2]): every one of the 9 commitments (11 G1 elements) of the proof recomputed
  on the CPU from the device-exported polynomials with the C restatement's Pippenger (oracle/c/ref_hotpath.c) plus the
  hiding part re-derived from the zk_rng stream -- "verifies" becomes "every commitment is the unique right point".
"""
import hashlib
import json
import os
import numpy as np
import pytest
from oracle import fields as F, curve as EC, fs as FS
from marlin_amd import marlin as GM

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LARGE = json.load(open(os.path.join(ROOT, "tests", "golden", "marlin_proofs_large.json")))
TAU, GAMMA, SEED = int(LARGE["tau"], 16), int(LARGE["gamma"], 16), bytes.fromhex(LARGE["zk_seed"])
BLS = pytest.mark.skipif(F.CURVE != "bls12_381", reason="fixtures and the C restatement are BLS12-381 / MarlinKZG10")


def _fr_bytes_of(arr):
    """(n,4) uint64 Montgomery -> concatenated canonical 32-byte LE encodings, trailing zero coefficients stripped
    (DensePolynomial::from_coefficients_vec)."""
    from oracle import cref
    a = np.ascontiguousarray(arr, dtype=np.uint64).copy()
    if len(a):
        cref.lib().ref_fr_from_mont(a.ctypes.data, len(a))
    nz = np.nonzero(a.any(axis=1))[0]
    a = a[: (nz[-1] + 1) if len(nz) else 0]
    return a.tobytes()


@BLS
@pytest.mark.parametrize("case", LARGE["cases"], ids=lambda c: "2^%d" % (c["num_constraints"].bit_length() - 1))
def test_proof_bytes_match_golden_large(gpu, case):
    n = case["num_constraints"]
    a, b = int(case["a"], 16), int(case["b"], 16)
    srs = GM.universal_setup(n, n, 3 * n, TAU, GAMMA)
    assert srs.max_degree == case["srs_max_degree"]
    ncp, ni, mats, inst, wit = GM.dummy_circuit(a, b, case["num_variables"], n)
    pk = GM.index(srs, ncp, ni, mats)
    assert (pk.H, pk.K) == (case["H"], case["K"])
    assert hashlib.blake2s(pk.vk_bytes()).hexdigest() == case["vk_bytes_blake2s"]
    proof = GM.prove(pk, inst, wit, SEED)
    # which polynomial diverged, if any (the proof comparison below is the assertion that counts)
    bad = [l for l, h in case["poly_blake2s"].items() if hashlib.blake2s(_fr_bytes_of(pk.get_poly(l))).hexdigest() != h]
    assert not bad, bad
    assert proof.hex() == case["proof_bytes"]


@BLS
@pytest.mark.parametrize("log_n", [16, 18])
def test_device_mask_polynomial_equals_sequential_stream(gpu, log_n):
    from tests import zkstream as ZS
    n = 1 << log_n
    srs = GM.universal_setup(n, n, 3 * n, TAU, GAMMA)
    ncp, ni, mats, inst, wit = GM.dummy_circuit(3, 5, 10, n)
    pk = GM.index(srs, ncp, ni, mats)
    GM.prove(pk, inst, wit, SEED)
    want = ZS.prove_zk_draws(SEED, n)["mask"]
    got = pk.get_poly("mask_poly")
    assert got.shape == want.shape == (3 * n, 4)
    assert np.array_equal(got, want), "first mismatch at coefficient %d" % int(np.nonzero((got != want).any(axis=1))[0][0])


@BLS
@pytest.mark.parametrize("log_n", [18, 20])
def test_every_commitment_pinned_by_cpu_msm(gpu, log_n):
    from oracle import cref
    from tests import zkstream as ZS
    from tests.util import limbs_to_fq
    from tests.verify_adapter import parse_proof
    rng = FS.test_rng()
    a, b = FS.fr_rand(rng), FS.fr_rand(rng)
    n = 1 << log_n
    srs = GM.universal_setup(n, n, 3 * n, TAU, GAMMA)
    ncp, ni, mats, inst, wit = GM.dummy_circuit(a, b, 10, n)
    pk = GM.index(srs, ncp, ni, mats)
    proof = parse_proof(GM.prove(pk, inst, wit, SEED))
    H, K, D = pk.H, pk.K, srs.max_degree
    bases = srs.powers_of_g.download()
    assert bases.shape == (D + 1, 2 * F.FQ_LIMBS64)
    threads = os.cpu_count() or 1
    zk = ZS.prove_zk_draws(SEED, H)
    gamma_pows = [EC.scalar_mul(EC.G1_GEN, GAMMA * pow(TAU, i, F.R_MOD) % F.R_MOD) for i in range(3)]

    def commit(label, offset, blind):
        coeffs = pk.get_poly(label)
        xyz = cref.msm(bases[offset:offset + len(coeffs)], coeffs, montgomery=True, threads=threads)
        xy, inf = cref.g1_to_affine(xyz)
        L = F.FQ_LIMBS64
        pt = None if inf else (limbs_to_fq(xy[:L]), limbs_to_fq(xy[L:]))
        if blind is not None:
            pt = EC.add(pt, EC.msm_naive(gamma_pows, blind))
        return pt

    c1, c2, c3 = proof.commitments
    want = {
        "w": (c1[0], None, zk["blind_w"], None), "z_a": (c1[1], None, zk["blind_za"], None), "z_b": (c1[2], None, zk["blind_zb"], None),
        "mask_poly": (c1[3], None, None, None), "t": (c2[0], None, None, None),
        "g_1": (c2[1], H - 2, zk["blind_g1"], zk["blind_g1_shifted"]), "h_1": (c2[2], None, None, None),
        "g_2": (c3[0], K - 2, None, None), "h_2": (c3[1], None, None, None),
    }
    bad = []
    for label, ((comm, shifted), bound, blind, sblind) in want.items():
        if commit(label, 0, blind) != comm:
            bad.append(label)
        if bound is None:
            assert shifted is None
        elif shifted is None or commit(label, D - bound, sblind) != shifted[0]:
            bad.append(label + " (shifted)")
    assert not bad, bad


@BLS
def test_gpu_proof_as_wire_bytes_verifies_under_the_pairing(gpu):
    """prove on the device at 2^12 -> CanonicalSerialize bytes (what a stock arkworks verifier would read) -> the
    oracle's Marlin::verify with the real pairing: accept / reject (src/test.rs:158-161)."""
    from tests.verify_adapter import oracle_verify
    rng = FS.test_rng()
    a, b = FS.fr_rand(rng), FS.fr_rand(rng)
    n = 1 << 12
    srs = GM.universal_setup(n, n, 3 * n, TAU, GAMMA)
    ncp, ni, mats, inst, wit = GM.dummy_circuit(a, b, 10, n)
    pk = GM.index(srs, ncp, ni, mats)
    flat = GM.prove(pk, inst, wit, SEED)
    wire = GM.proof_serialize(flat)
    assert len(wire) == 855 and GM.proof_deserialize(wire) == flat
    vk = pk.vk_bytes()
    c = a * b % F.R_MOD
    assert oracle_verify(vk, srs.max_degree, TAU, GAMMA, [c], wire, use_pairing=True, wire=True)
    assert not oracle_verify(vk, srs.max_degree, TAU, GAMMA, [a], wire, use_pairing=True, wire=True)
