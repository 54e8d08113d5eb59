"""Proof wire format + pairing verifier, CPU only.

* `CanonicalSerialize for Proof` (src/data_structures.rs:100-110): the product's host-side serialiser
  (mh_marlin_proof_serialize, marlin_amd/csrc/wire_host.h -- no GPU involved) against the oracle's restatement
  (oracle/marlin.py proof_serialize) on every golden proof; round trip through the validating deserialisers.
* `Marlin::verify` (src/lib.rs:315-433) with the real BLS12-381 pairing (oracle/pairing.py) instead of the known-tau
  shortcut: accepts the golden proof, rejects a wrong public input and a tampered proof (src/test.rs:158-161).
"""
import json
import os
import pytest
from oracle import ahp as AHP, marlin as MR, fs as FS, fields as F
from marlin_amd import marlin as GM
from tests.verify_adapter import parse_proof

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "marlin_proofs.json")))
pytestmark = pytest.mark.skipif(F.CURVE != "bls12_381", reason="golden fixtures are BLS12-381 / MarlinKZG10")


@pytest.mark.parametrize("case", GOLD["cases"], ids=lambda c: "%s-%d-%d" % (c["kind"], c["num_constraints"], c["num_variables"]))
def test_canonical_serialize_matches_oracle_and_round_trips(case):
    flat = bytes.fromhex(case["proof_bytes"])
    wire = GM.proof_serialize(flat)
    assert wire == MR.proof_serialize(parse_proof(flat))
    # 9 commitments: 9 x (48 + 1) + 2 x 48 shifted; lengths 8 + 3 x 8; evaluations 8 + 4 x 32; prover messages 8 + 3;
    # openings 8 + (48 + 1 + 32) + (48 + 1) + 1   (random_v: Some at beta, None at gamma)
    assert len(wire) == 855
    assert GM.proof_deserialize(wire) == flat
    assert MR.proof_bytes(MR.proof_deserialize(wire)) == flat


def test_deserialize_rejects_invalid_encodings():
    flat = bytes.fromhex(GOLD["cases"][0]["proof_bytes"])
    wire = bytearray(GM.proof_serialize(flat))
    first_g1 = 16                                    # after the two u64 lengths
    # y-sign flag flipped: still a valid encoding, of the negated point
    w2 = bytearray(wire); w2[first_g1 + 47] ^= 0x80
    other = GM.proof_deserialize(bytes(w2))
    assert other != flat and other[:48] == flat[:48] and other[48:96] != flat[48:96]
    bad = []
    w3 = bytearray(wire); w3[first_g1] ^= 1; bad.append(w3)                     # x no longer on the curve (with overwhelming probability) or off-subgroup
    w4 = bytearray(wire); w4[first_g1 + 47] |= 0x40; bad.append(w4)             # infinity flag with x != 0
    w5 = bytearray(wire); w5[0] = 4; bad.append(w5)                             # wrong outer length
    bad.append(wire[:-1]); bad.append(wire + b"\x00")
    w6 = bytearray(wire); w6[-1] = 1; bad.append(w6)                            # BatchLCProof.evals must be None
    for w in bad:
        with pytest.raises(GM._lib.MarlinHipError):
            GM.proof_deserialize(bytes(w))
        with pytest.raises(AssertionError):
            MR.proof_deserialize(bytes(w))
    # an x that IS on the curve but not in the prime-order subgroup is rejected too (cofactor of BLS12-381 G1 != 1)
    x = 1
    while True:
        y2 = (x ** 3 + F.G1_B) % F.Q_MOD
        y = pow(y2, (F.Q_MOD + 1) // 4, F.Q_MOD)
        if y * y % F.Q_MOD == y2:
            from oracle import curve as EC
            if EC.add(EC.scalar_mul((x, y), F.R_MOD - 1), (x, y)) is not None:      # [r]P != O
                break
        x += 1
    w7 = bytearray(wire); w7[first_g1:first_g1 + 48] = x.to_bytes(48, "little")
    with pytest.raises(GM._lib.MarlinHipError):
        GM.proof_deserialize(bytes(w7))


def test_verify_with_real_pairing_accepts_and_rejects():
    case = GOLD["cases"][0]                            # test.rs shape (25, 25)
    a, b = int(case["a"], 16), int(case["b"], 16)
    nc, nv = case["num_constraints"], case["num_variables"]
    cs = AHP.pad_and_square(AHP.finalize_test_circuit(AHP.test_circuit(a, b, nc, nv)))
    tau, gamma = int(GOLD["tau"], 16), int(GOLD["gamma"], 16)
    srs = MR.universal_setup(nc, nv, 3 * max(nc, nv), tau, gamma)
    pk = MR.marlin_index(srs, cs)
    # the proof travels as wire bytes, like it would to a stock verifier
    pr = MR.proof_deserialize(GM.proof_serialize(bytes.fromhex(case["proof_bytes"])))
    c = a * b % F.R_MOD
    d = c * b % F.R_MOD
    assert MR.verify(pk, [c, d], pr, use_pairing=True)
    assert not MR.verify(pk, [a, a], pr, use_pairing=True)
    pr.evaluations[2] = (pr.evaluations[2] + 1) % F.R_MOD
    assert not MR.verify(pk, [c, d], pr, use_pairing=True)


def test_serialized_size_follows_from_the_proof_structure():
    """`Proof::serialized_size` (src/data_structures.rs:129-161 prints it) by the CanonicalSerialize rules of ark-serialize
    0.3 [UPSTREAM-RECALLED]: u64 length prefix per Vec, 48-byte compressed G1, one tag byte per Option, 32-byte Fr.
    MarlinKZG10: commitments 8 + 3 x 8 + 7 x (48 + 1) + 2 x (48 + 1 + 48) = 569; evaluations 8 + 4 x 32 = 136;
    prover_messages 8 + 3 x 1 = 11; pc_proof 8 + (48 + 1 + 32) + (48 + 1) + 1 = 139  ->  855.
    SonicKZG10: commitments 8 + 3 x 8 + 9 x 48 = 464  ->  750.
    (README.md:87-88 quotes 880 / 784 bytes: those belong to an earlier proof structure of the paper's time, with more
    evaluations; they do not describe `Proof` as src/data_structures.rs:100-110 defines it.)"""
    marlin = (8 + 3 * 8 + 7 * (48 + 1) + 2 * (48 + 1 + 48)) + (8 + 4 * 32) + (8 + 3) + (8 + (48 + 1 + 32) + (48 + 1) + 1)
    sonic = (8 + 3 * 8 + 9 * 48) + (8 + 4 * 32) + (8 + 3) + (8 + (48 + 1 + 32) + (48 + 1) + 1)
    assert (marlin, sonic) == (855, 750)
    # the flat ToBytes layout (uncompressed points with infinity bytes) of the same proofs
    assert GM.proof_bytes_len("marlin") == 9 * 195 + 4 * 32 + 2 * (97 + 1 + 32) and GM.proof_bytes_len("sonic") == 9 * 97 + 4 * 32 + 2 * (97 + 1 + 32)
