"""CPU-only: the large-size recomputation of the PC layer (tests/cpu_open.py, C restatement underneath) reproduces the
pure-Python oracle's proof -- commitments, evaluations, both openings -- from the oracle's own polynomials at a size
both can run, for MarlinKZG10 and SonicKZG10 on the curve the process selected.  This is what licenses its use as the
checker at 2^18 .. 2^22 in tests/test_gpu_parity_pins.py."""
import numpy as np
import pytest
from oracle import fields as F, ahp as AHP, marlin as MR, fs as FS
from tests import cpu_open as CO, zkstream as ZS
from tests.util import fr_to_np, points_to_np

TAU, GAMMA, SEED = 0x1b2c3d4e5f60718293a4b5c6d7e8f901, 0x2468ace13579bdf, bytes(range(7, 39))


@pytest.mark.parametrize("pc", ["marlin", "sonic"])
def test_cpu_recomputation_equals_oracle_proof(pc):
    a, b, n = 0x1234567, 0x7654321, 32
    cs = AHP.pad_and_square(AHP.dummy_circuit(a, b, 10, n))
    srs = MR.universal_setup(n, n, 3 * n, TAU, GAMMA)
    pk = MR.marlin_index(srs, cs, pc)
    pr = MR.prove(pk, cs, FS.ChaChaRng(SEED, 20))
    flat = MR.proof_bytes(pr)
    H, K = pk.index.domain_h.size, pk.index.domain_k.size
    polys = {l: fr_to_np(p) for l, p in pr.polys.items()}
    cpc = CO.CpuPC(points_to_np(srs.powers_of_g), srs.max_degree, TAU, GAMMA, threads=2)
    zk = ZS.prove_zk_draws(SEED, H)
    got = CO.recompute_proof(cpc, polys, zk, MR.vk_bytes(pk), [a * b % F.R_MOD], flat, H, K, pc)
    assert got.commitments == pr.commitments
    assert got.evaluations == pr.evaluations
    assert got.pc_proof == pr.pc_proof
    assert MR.proof_bytes(got) == flat
