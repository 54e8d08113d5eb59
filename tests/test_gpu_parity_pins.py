"""Parity pins above the sizes the pure-Python oracle can prove at.

* byte-identical proofs against oracle-generated golden fixtures at 2^10 (BASELINE configs[0]), 2^12 and 2^14
  (tests/golden/marlin_proofs_large.json, tests/golden/make_golden.py large: the pure-Python oracle) and at 2^16, 2^18
  (configs[1]) and 2^20 (configs[2]) (tests/golden/marlin_proofs_xl.json, make_golden.py xl: the same oracle with its C
  backend for NTT / MSM / SRS powers, oracle/accel.py) -- every prover polynomial hashed, the proof compared byte for byte;
* the device's `DensePolynomial::rand` (rng.cuh: ChaCha blocks in parallel + rejection sampling as stream
  compaction) against the sequential `Fp256::rand` stream at 2^16 / 2^18 (3|H| draws, prover.rs:370-380);
* at 2^18 (configs[1]), 2^20 (configs[2]) and, in the BN254 subprocess, BN254 +
  SonicKZG10 at 2^20 (configs[4]): the WHOLE proof -- 9 commitments (11 G1 elements), 4 evaluations, both opening
  proofs W_beta / W_gamma / random_v -- recomputed on the CPU from the device-exported polynomials the way the
  reference computes them (tests/cpu_open.py: one MSM per KZG10::commit / witness with the C restatement's Pippenger,
  linear combinations and synthetic division on the CPU, hiding parts re-derived from the zk_rng stream) and compared
  byte for byte -- "verifies" becomes "every group element and field element of the proof is the unique right one".
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
# 2^16, 2^18 (BASELINE configs[1]), 2^20 (configs[2]): whole proofs from the oracle with its C backend for NTT / MSM / SRS
# (oracle/accel.py: a CPU prover that shares no code with the device; tests/golden/make_golden.py xl)
_XL_PATH = os.path.join(ROOT, "tests", "golden", "marlin_proofs_xl.json")
XL = json.load(open(_XL_PATH)) if os.path.exists(_XL_PATH) else {"cases": []}
assert not XL["cases"] or (XL["tau"], XL["gamma"], XL["zk_seed"]) == (LARGE["tau"], LARGE["gamma"], LARGE["zk_seed"])
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
@pytest.mark.parametrize("case", LARGE["cases"] + XL["cases"], ids=lambda c: "2^%d" % (c["num_constraints"].bit_length() - 1))
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


# the other configurations (make_golden.py xl-cfg): this process's curve with SonicKZG10 (benches/bench.rs's own shape at 2^16;
# 2^20 = BASELINE configs[4] on BN254) and, on BN254, MarlinKZG10.  The BN254 files run in tests/test_gpu_bn254.py's subprocess.
def _cfg_cases():
    out = []
    for pc in ("sonic", "marlin"):
        path = os.path.join(ROOT, "tests", "golden", "marlin_proofs_xl_%s_%s.json" % (F.CURVE, pc))
        if os.path.exists(path):
            g = json.load(open(path))
            assert (g["curve"], g["pc"], g["tau"], g["gamma"], g["zk_seed"]) == (F.CURVE, pc, LARGE["tau"], LARGE["gamma"], LARGE["zk_seed"])
            out += [(pc, c) for c in g["cases"]]
    return out


@pytest.mark.parametrize("pc,case", _cfg_cases(), ids=lambda v: v if isinstance(v, str) else "2^%d" % (v["num_constraints"].bit_length() - 1))
def test_proof_bytes_match_golden_xl_cfg(gpu, pc, case):
    """Whole proofs of the CPU oracle (C backend for NTT / MSM / SRS) for the other PC scheme / the other curve, every prover
    polynomial hashed: the device's proof is the same bytes."""
    n = case["num_constraints"]
    a, b = int(case["a"], 16), int(case["b"], 16)
    srs = GM.universal_setup(n, n, 3 * n, TAU, GAMMA, pc=pc)
    assert srs.max_degree == case["srs_max_degree"]
    ncp, ni, mats, inst, wit = GM.dummy_circuit(a, b, case["num_variables"], n)
    pk = GM.index(srs, ncp, ni, mats, pc=pc)
    assert (pk.H, pk.K) == (case["H"], case["K"])
    assert hashlib.blake2s(pk.vk_bytes()).hexdigest() == case["vk_bytes_blake2s"]
    proof = GM.prove(pk, inst, wit, SEED)
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


ALL_LABELS = ["w", "z_a", "z_b", "mask_poly", "t", "g_1", "h_1", "g_2", "h_2", "row", "col", "a_val", "b_val", "c_val", "row_col"]


def _whole_proof_pinned(log_n, pc):
    """prove on the device, export the 15 polynomials, recompute on the CPU -- the reference's way, one MSM per
    KZG10::commit / witness, tests/cpu_open.py -- all 9 commitments (11 G1 elements for MarlinKZG10), the 4 evaluations
    and BOTH opening proofs (W_beta, W_gamma, random_v): the flat proof bytes must be identical."""
    from oracle import cref, marlin as MR
    from tests import zkstream as ZS, cpu_open as CO
    rng = FS.test_rng()
    a, b = FS.fr_rand(rng), FS.fr_rand(rng)
    n = 1 << log_n
    srs = GM.universal_setup(n, n, 3 * n, TAU, GAMMA, pc=pc)
    ncp, ni, mats, inst, wit = GM.dummy_circuit(a, b, 10, n)
    pk = GM.index(srs, ncp, ni, mats, pc=pc)
    flat = GM.prove(pk, inst, wit, SEED)
    H, K, D = pk.H, pk.K, srs.max_degree
    bases = srs.powers_of_g.download()
    assert bases.shape == (D + 1, 2 * F.FQ_LIMBS64)
    threads = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cpc = CO.CpuPC(bases, D, TAU, GAMMA, threads)
    polys = {l: pk.get_poly(l) for l in ALL_LABELS}
    got = CO.recompute_proof(cpc, polys, ZS.prove_zk_draws(SEED, H), pk.vk_bytes(), [a * b % F.R_MOD], flat, H, K, pc)
    want = parse_proof_flat(flat, pc)
    bad = [(r, i) for r in range(3) for i in range(len(want.commitments[r])) if got.commitments[r][i] != want.commitments[r][i]]
    assert not bad, "commitments (round, index) differ: %r" % bad
    assert got.evaluations == want.evaluations
    assert got.pc_proof[0] == want.pc_proof[0], "opening at beta (W, random_v)"
    assert got.pc_proof[1] == want.pc_proof[1], "opening at gamma (W, random_v)"
    assert MR.proof_bytes(got) == flat


def parse_proof_flat(flat, pc):
    from tests.verify_adapter import parse_proof
    return parse_proof(flat, pc)


@BLS
@pytest.mark.parametrize("log_n", [18, 20])
def test_whole_proof_pinned_by_cpu_recomputation(gpu, log_n):
    """BASELINE configs[1] and [2]: MarlinKZG10 on BLS12-381.  configs[3]'s size on one GPU (2^22) is pinned byte for byte by
    test_proof_bytes_match_golden_large[2^22]: every commitment, evaluation and opening of the independent CPU prover's proof
    (oracle/accel.py; tests/golden/marlin_proofs_xl.json) and the hash of every prover polynomial -- the same CPU Pippenger that
    this recomputation would run for 124 s more of the suite's budget."""
    _whole_proof_pinned(log_n, "marlin")


@pytest.mark.skipif(F.CURVE != "bn254", reason="BASELINE configs[4] is BN254; runs in tests/test_gpu_bn254.py's subprocess")
def test_whole_proof_pinned_bn254_sonic_2p20(gpu):
    """BASELINE configs[4]: BN254 + SonicKZG10 at 2^20 constraints, against libref_hotpath_bn254.so."""
    _whole_proof_pinned(20, "sonic")


@BLS
def test_whole_proof_pinned_sonic_2p16_reference_bench_shape(gpu):
    """benches/bench.rs:75-83's own shape (2^16 constraints, SonicKZG10, BLS12-381)."""
    _whole_proof_pinned(16, "sonic")


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
