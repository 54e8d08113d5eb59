"""Consumer of arkworks-produced golden vectors.

`shim/tests/parity.rs::write_self_contained_golden_vectors` -- run with `cargo test` on a machine that has Rust, which
this image has not -- writes `arkworks_golden.json`: per case the SRS (known tau / gamma on the standard generators, for
the small cases also the compressed points), the circuit parameters, the zk seed, and the bytes the STOCK arkworks stack
produced: `to_bytes![index_vk]` and `proof.serialize(..)`.  Dropped into tests/golden/ as `arkworks_*.json`, every case is
replayed here
  * through the Python oracle (CPU; cases up to 2^10 constraints), and
  * through `mh_marlin_index` / `mh_marlin_prove` on the device (all cases; SRS through `mh_bases_upload_serialized`
    where the fixture embeds it, regenerated with `mh_srs_powers` and checked against the fixture's hash otherwise),
and the bytes must be equal.  That is the test that turns "parity unpinned" into "pinned against arkworks".

No such file can be produced in this image, so the `arkworks` parametrisations SKIP with that reason.  The same code runs
on `oracle_in_arkworks_format.json` -- the identical format written by this repository's oracle
(tests/golden/make_arkworks_format_fixture.py) -- which pins nothing about arkworks but proves the consumer works.
"""
import glob
import hashlib
import json
import os
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
NO_ARKWORKS = ("no arkworks-produced fixture (tests/golden/arkworks_*.json): it is written by `cargo test` of shim/ "
               "(shim/tests/parity.rs::write_self_contained_golden_vectors) and this image has no Rust toolchain")


def _cases(pattern):
    out = []
    for path in sorted(glob.glob(os.path.join(GOLD, pattern))):
        doc = json.load(open(path))
        assert doc["format"] == 1
        for c in doc["cases"]:
            out.append(pytest.param(c, id="%s:%s:%s" % (os.path.basename(path).split(".")[0], c["name"], c["pc"])))
    return out


ARKWORKS = _cases("arkworks_*.json") or [pytest.param(None, id="absent", marks=pytest.mark.skip(reason=NO_ARKWORKS))]
SELFCHECK = _cases("oracle_in_arkworks_format.json")


def _int(h):
    return int.from_bytes(bytes.fromhex(h), "little")


def _oracle_replay(c):
    from oracle import fields as F, ahp as AHP, marlin as MR, fs as FS
    assert F.CURVE == c["curve"]
    ci, s = c["circuit"], c["srs"]
    a, b = _int(ci["a"]), _int(ci["b"])
    if ci["kind"] == "test":
        cs = AHP.finalize_test_circuit(AHP.test_circuit(a, b, ci["num_constraints"], ci["num_variables"]))
    else:
        cs = AHP.dummy_circuit(a, b, ci["num_variables"], ci["num_constraints"])
    cs = AHP.pad_and_square(cs)
    srs = MR.universal_setup(s["num_constraints"], s["num_variables"], s["num_non_zero"], _int(s["tau"]), _int(s["gamma"]))
    assert srs.max_degree == s["max_degree"]
    g_bytes = b"".join(MR.g1_compressed(p) for p in srs.powers_of_g)
    assert hashlib.blake2s(g_bytes).hexdigest() == s["powers_of_g_blake2s"], "SRS differs: generator or tau convention"
    if "powers_of_g" in s:
        assert g_bytes.hex() == s["powers_of_g"]
    pk = MR.marlin_index(srs, cs, c["pc"])
    assert MR.vk_bytes(pk).hex() == c["vk_to_bytes"], "IndexVerifierKey bytes differ"
    pr = MR.prove(pk, cs, FS.ChaChaRng(bytes.fromhex(c["zk_seed"]), c["zk_rounds"]))
    pub = [x.to_bytes(32, "little").hex() for x in AHP.public_input(AHP.prover_init(pk.index, cs))]   # padded (lib.rs:323-333)
    n = len(c["public_input"])
    assert pub[:n] == c["public_input"] and all(int(x, 16) == 0 for x in pub[n:])
    assert MR.proof_serialize(pr).hex() == c["proof"], "proof bytes differ"


def _device_replay(c):
    import marlin_amd as M
    from marlin_amd import marlin as GM
    from marlin_amd.api import Bases
    ci, s = c["circuit"], c["srs"]
    a, b = _int(ci["a"]), _int(ci["b"])
    D = s["max_degree"]
    srs = GM.universal_setup(s["num_constraints"], s["num_variables"], s["num_non_zero"], _int(s["tau"]), _int(s["gamma"]), pc=c["pc"])
    assert srs.max_degree == D
    if "powers_of_g" in s:
        # the fixture's own points, decoded and validated on the device: what a host that READS an arkworks SRS file does
        embedded = Bases.from_serialized(bytes.fromhex(s["powers_of_g"]), D + 1, compressed=True)
        assert np.array_equal(embedded.download(), srs.powers_of_g.download()), "device SRS from tau != fixture's points"
        srs.powers_of_g.free()
        srs.powers_of_g = embedded
        ng = len(s["powers_of_gamma_g"]) // 2 // (8 * M._lib.FQ_LIMBS)
        gam = Bases.from_serialized(bytes.fromhex(s["powers_of_gamma_g"]), ng, compressed=True)
        assert np.array_equal(gam.download(0, 3), srs.powers_of_gamma_g.download(0, 3))
        if c["pc"] == "sonic":
            srs.powers_of_gamma_g.free()
            srs.powers_of_gamma_g = gam
    else:
        from oracle import marlin as MR
        from tests.util import limbs_to_fq
        L = M._lib.FQ_LIMBS
        h = hashlib.blake2s()
        pts = srs.powers_of_g.download()
        for row in pts:                                    # compressed image of the device-generated SRS
            h.update(MR.g1_compressed((limbs_to_fq(row[:L]), limbs_to_fq(row[L:]))))
        assert h.hexdigest() == s["powers_of_g_blake2s"], "SRS differs: generator or tau convention"
    if ci["kind"] == "test":
        ncp, ni, mats, inst, wit = GM.test_circuit(a, b, ci["num_constraints"], ci["num_variables"])
    else:
        ncp, ni, mats, inst, wit = GM.dummy_circuit(a, b, ci["num_variables"], ci["num_constraints"])
    pk = GM.index(srs, ncp, ni, mats, pc=c["pc"])
    assert pk.vk_bytes().hex() == c["vk_to_bytes"], "IndexVerifierKey bytes differ"
    flat = GM.prove(pk, inst, wit, bytes.fromhex(c["zk_seed"]), c["zk_rounds"])
    assert GM.proof_serialize(flat, pc=c["pc"]).hex() == c["proof"], "proof bytes differ"


def _small(c):
    return c["circuit"]["num_constraints"] <= 1 << 10 and c["circuit"]["num_variables"] <= 1 << 10


@pytest.mark.parametrize("case", ARKWORKS)
def test_oracle_reproduces_arkworks_bytes(case):
    if not _small(case):
        pytest.skip("beyond what the pure-Python oracle proves in minutes; the device replay covers it")
    _oracle_replay(case)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ARKWORKS)
def test_device_reproduces_arkworks_bytes(gpu, case):
    _device_replay(case)


@pytest.mark.parametrize("case", SELFCHECK)
def test_consumer_selfcheck_oracle(case):
    """the consumer itself, on a fixture of the same format made by this repository's oracle (pins nothing upstream)"""
    _oracle_replay(case)


@pytest.mark.gpu
@pytest.mark.parametrize("case", SELFCHECK)
def test_consumer_selfcheck_device(gpu, case):
    _device_replay(case)
