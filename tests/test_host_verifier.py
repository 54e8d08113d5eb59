"""The product's host-side verifier (mh_marlin_verify) and pairing (mh_pairing_product_is_one) -- no GPU involved --
against the oracle.  Marlin::verify (/root/reference src/lib.rs:315-433) must accept the golden proofs and reject a wrong
public input (src/test.rs:158-161), and reject tampered evaluations / commitments / openings; the pairing must be
bilinear, non-degenerate and take the decisions oracle/pairing.py takes."""
import json
import os
import random
import numpy as np
import pytest
from oracle import ahp as AHP, marlin as MR, fs as FS, fields as F, curve as EC
from marlin_amd import marlin as GM
from tests.util import fr_to_np, fq_to_limbs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = F.R_MOD
BLS = F.CURVE == "bls12_381"


def g1_np(pt):
    return np.array(fq_to_limbs(pt[0]) + fq_to_limbs(pt[1]), dtype=np.uint64)


def g2_np(pt):
    (x0, x1), (y0, y1) = pt
    return np.array(fq_to_limbs(x0) + fq_to_limbs(x1) + fq_to_limbs(y0) + fq_to_limbs(y1), dtype=np.uint64)


def test_pairing_product_bilinear_and_matches_oracle():
    from oracle import pairing as PR
    rng = random.Random(5)
    a, b = rng.randrange(1, R), rng.randrange(1, R)
    P, Q = EC.G1_GEN, PR.G2_GEN
    aP, bQ, aQ = EC.scalar_mul(P, a), PR.g2_mul(Q, b), PR.g2_mul(Q, a)
    abP = EC.scalar_mul(P, a * b % R)
    cases = [
        ([(aP, bQ), (EC.neg(abP), Q)], True),          # e(aP, bQ) = e(abP, Q)
        ([(aP, Q), (EC.neg(P), aQ)], True),            # e(aP, Q) = e(P, aQ)
        ([(aP, Q), (EC.neg(P), bQ)], False),
        ([(P, Q)], False),                             # non-degenerate
    ]
    for pairs, want in cases:
        got = GM.pairing_product_is_one([g1_np(p) for p, _ in pairs], [g2_np(q) for _, q in pairs])
        assert got == want
    assert PR.pairing_product_is_one(cases[0][0]) and not PR.pairing_product_is_one(cases[2][0])
    bad = g2_np(Q).copy()
    bad[0] ^= 1
    with pytest.raises(Exception, match="curve"):
        GM.pairing_product_is_one([g1_np(P)], [bad])


def _case(kind, nc, nv):
    rng = FS.test_rng()
    a, b = FS.fr_rand(rng), FS.fr_rand(rng)
    if kind == "test_circuit":
        cs = AHP.finalize_test_circuit(AHP.test_circuit(a, b, nc, nv))
        pub = [a * b % R, a * b % R * b % R]
    else:
        cs = AHP.dummy_circuit(a, b, nv, nc)
        pub = [a * b % R]
    return a, AHP.pad_and_square(cs), pub


def _vk_elements(srs, pk, pc="marlin"):
    from oracle import pairing as PR
    H, K = pk.index.domain_h.size, pk.index.domain_k.size
    shifts = [srs.max_degree - (H - 2), srs.max_degree - (K - 2)]
    if pc == "sonic":
        sp = [g2_np(PR.g2_mul(PR.G2_GEN, pow(srs.tau, -d, R))) for d in shifts]
    else:
        sp = [g1_np(srs.powers_of_g[d]) for d in shifts]
    return [g1_np(srs.g), g1_np(srs.gamma_g), g2_np(PR.G2_GEN), g2_np(PR.g2_mul(PR.G2_GEN, srs.tau))] + sp


GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "marlin_proofs.json")))
PICK = [c for c in GOLD["cases"] if (c["kind"], c["num_constraints"], c["num_variables"]) in
        (("test_circuit", 25, 25), ("test_circuit", 100, 25), ("dummy_circuit", 64, 10))]


@pytest.mark.skipif(not BLS, reason="golden fixtures are BLS12-381 / MarlinKZG10")
@pytest.mark.parametrize("case", PICK, ids=lambda c: "%s-%d-%d" % (c["kind"], c["num_constraints"], c["num_variables"]))
def test_host_verifier_accepts_golden_and_rejects_tampering(case):
    from oracle import pairing as PR
    tau, gamma = int(GOLD["tau"], 16), int(GOLD["gamma"], 16)
    nc, nv = case["num_constraints"], case["num_variables"]
    a, cs, pub = _case(case["kind"], nc, nv)
    srs = MR.universal_setup(max(nc, nv), max(nc, nv), 3 * max(nc, nv), tau, gamma)
    pk = MR.marlin_index(srs, cs)
    vkb = MR.vk_bytes(pk)
    proof = bytes.fromhex(case["proof_bytes"])
    els = _vk_elements(srs, pk)

    assert GM.verify(vkb, *els, fr_to_np(pub), proof)
    assert not GM.verify(vkb, *els, fr_to_np([a] * len(pub)), proof)                 # src/test.rs:161
    assert not GM.verify(bytes([vkb[0] ^ 1]) + vkb[1:], *els, fr_to_np(pub), proof)    # another index_info: other challenges
    wrong_beta = list(els)
    wrong_beta[3] = g2_np(PR.g2_mul(PR.G2_GEN, srs.tau + 1))
    assert not GM.verify(vkb, *wrong_beta, fr_to_np(pub), proof)

    # layout of the flat proof (oracle.marlin.proof_bytes): commitments | 4 evaluations | 2 openings
    fq = F.FQ_BYTES
    ev0 = len(proof) - 4 * 32 - 2 * (2 * fq + 1 + 1 + 32)
    t = bytearray(proof)
    t[ev0] ^= 1                                                                      # an evaluation
    assert not GM.verify(vkb, *els, fr_to_np(pub), bytes(t))
    t = bytearray(proof)
    t[len(proof) - 2 * (2 * fq + 1 + 1 + 32) + 2 * fq + 2] ^= 1                      # random_v of the opening at beta
    assert not GM.verify(vkb, *els, fr_to_np(pub), bytes(t))
    t = bytearray(proof)
    t[3] ^= 1                                                                        # x of the first commitment
    with pytest.raises(Exception):
        GM.verify(vkb, *els, fr_to_np(pub), bytes(t))
    # swap two commitments (both valid points): the transcript and the combination change
    c = 2 * (2 * fq + 1) + 1
    t = bytearray(proof)
    t[0:c], t[c:2 * c] = proof[c:2 * c], proof[0:c]
    assert not GM.verify(vkb, *els, fr_to_np(pub), bytes(t))


@pytest.mark.parametrize("pc", ["marlin", "sonic"])
def test_host_verifier_agrees_with_oracle_pairing_verifier(pc):
    """The oracle's verify(use_pairing=True) -- real pairing, no tau -- and the product's verifier decide alike on a fresh
    proof (not a golden one) and on its corruptions, for both PC schemes, on the curve of this process."""
    from oracle import pairing as PR
    a, cs, pub = _case("test_circuit", 25, 25)
    srs = MR.universal_setup(25, 25, 75, 0xabcdef12345, 0x777)
    pk = MR.marlin_index(srs, cs, pc=pc)
    pr = MR.prove(pk, cs, FS.ChaChaRng(bytes(range(1, 33)), 20))
    els = _vk_elements(srs, pk, pc)
    vkb = MR.vk_bytes(pk)
    assert MR.verify(pk, pub, pr, use_pairing=True) and MR.verify(pk, pub, pr)
    assert GM.verify(vkb, *els, fr_to_np(pub), MR.proof_bytes(pr), pc=pc)
    assert not GM.verify(vkb, *els, fr_to_np([a] * len(pub)), MR.proof_bytes(pr), pc=pc)
    # the other scheme's layout is refused outright
    with pytest.raises(Exception, match="scheme"):
        GM.verify(vkb, *els[:4], *_vk_elements(srs, pk, "sonic" if pc == "marlin" else "marlin")[4:], fr_to_np(pub), MR.proof_bytes(pr),
                  pc="sonic" if pc == "marlin" else "marlin")
    # a wrong shift power breaks the degree-bound check only
    wrong = list(els)
    wrong[4] = g2_np(PR.g2_mul(PR.G2_GEN, 12345)) if pc == "sonic" else g1_np(EC.scalar_mul(EC.G1_GEN, 12345))
    assert not GM.verify(vkb, *wrong, fr_to_np(pub), MR.proof_bytes(pr), pc=pc)
    for k in range(4):
        saved = pr.evaluations[k]
        pr.evaluations[k] = (saved + 1) % R
        assert not MR.verify(pk, pub, pr, use_pairing=True)
        assert not GM.verify(vkb, *els, fr_to_np(pub), MR.proof_bytes(pr), pc=pc)
        pr.evaluations[k] = saved
    assert GM.verify(vkb, *els, fr_to_np(pub), MR.proof_bytes(pr), pc=pc)


def test_verifier_key_g2_elements_must_be_in_the_subgroup():
    """E'(Fq2) has a cofactor on both curves; a point of the twist outside G2 is refused."""
    from oracle import g2 as G2
    import itertools
    P = F.Q_MOD
    # a point of the twist found by trial x: outside G2 with overwhelming probability (checked below)
    for x0 in itertools.count(1):
        x = (x0, 1)
        rhs = G2.f2_add(G2.f2_mul(G2.f2_mul(x, x), x), G2.G2_B)
        # square root in Fq2 = Fq[u]/(u^2 + 1), p = 3 mod 4, through the norm: (y0 + y1 u)^2 = rhs
        y = None
        n = (rhs[0] * rhs[0] + rhs[1] * rhs[1]) % P
        if pow(n, (P - 1) // 2, P) != 1:
            continue
        assert P % 4 == 3
        sn = pow(n, (P + 1) // 4, P)
        for s in (sn, P - sn):
            t = (rhs[0] + s) * pow(2, -1, P) % P
            if pow(t, (P - 1) // 2, P) == 1:
                y0 = pow(t, (P + 1) // 4, P)
                y1 = rhs[1] * pow(2 * y0, -1, P) % P
                if G2.f2_mul((y0, y1), (y0, y1)) == rhs:
                    y = (y0, y1)
                    break
        if y is not None:
            pt = (x, y)
            break
    assert G2.g2_is_on_curve(pt) and G2.g2_add(G2.g2_mul(pt, R - 1), pt) is not None        # [r]pt != O: not in G2
    a, cs, pub = _case("test_circuit", 25, 25)
    srs = MR.universal_setup(25, 25, 75, 0xabcdef12345, 0x777)
    pk = MR.marlin_index(srs, cs)
    els = _vk_elements(srs, pk)
    els[2] = g2_np(pt)
    with pytest.raises(Exception, match="subgroup"):
        GM.verify(MR.vk_bytes(pk), *els, fr_to_np(pub), bytes(GM.proof_bytes_len("marlin")))


@pytest.mark.skipif(not BLS, reason="runs the file once more for the second curve")
def test_second_curve_in_a_subprocess():
    """BN254 (BASELINE configs[4]): same tests against libmarlin_hip_bn254.so and the oracle restated over BN254."""
    import subprocess
    import sys
    env = dict(os.environ, MARLIN_AMD_CURVE="bn254", ORACLE_CURVE="bn254")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", os.path.abspath(__file__)], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout
