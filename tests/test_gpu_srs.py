"""Device SRS generation ([tau^i]G) and the known-tau KZG identity commit(p) == [p(tau)]G
(SURVEY.md §8c item 2): an O(1) check of an MSM of any size."""
import numpy as np
import pytest
from oracle import fields as F, curve as EC, poly as OP
from tests.util import fr_to_np, jac_np_to_affine, rand_fr, limbs_to_fq

pytestmark = pytest.mark.gpu
TAU = 0x1f3a9c5d7e2b4a6f8091a2b3c4d5e6f708192a3b4c5d6e7f


def test_srs_powers_match_oracle(gpu):
    n = 40
    B = gpu.Bases.srs_powers(fr_to_np([TAU])[0], n)
    got = B.download()
    want = EC.srs_powers(TAU, n)
    for i in range(n):
        assert (limbs_to_fq(got[i, :F.FQ_LIMBS64]), limbs_to_fq(got[i, F.FQ_LIMBS64:])) == want[i]
    gamma = 0x5eed5eed5eed
    Bg = gpu.Bases.srs_powers(fr_to_np([TAU])[0], 5, scale_mont=fr_to_np([gamma])[0])
    got = Bg.download()
    for i in range(5):
        assert (limbs_to_fq(got[i, :F.FQ_LIMBS64]), limbs_to_fq(got[i, F.FQ_LIMBS64:])) == EC.scalar_mul(EC.G1_GEN, gamma * pow(TAU, i, F.R_MOD))


@pytest.mark.parametrize("log_n", [6, 10, 13])
def test_commit_equals_p_of_tau_small(gpu, log_n):
    n = 1 << log_n
    B = gpu.Bases.srs_powers(fr_to_np([TAU])[0], n)
    coeffs = rand_fr(n, log_n)
    out = gpu.msm(B, fr_to_np(coeffs))
    assert jac_np_to_affine(out) == EC.scalar_mul(EC.G1_GEN, OP.poly_eval(coeffs, TAU))


@pytest.mark.parametrize("log_n", [16, 20, 22])
def test_commit_full_size_properties(gpu, log_n):
    """BASELINE.json sizes (up to the 2^22-point SRS of the 2^20-constraint config):
    (1) a sparse polynomial's commitment equals [p(tau)]G exactly (checks far-apart SRS powers),
    (2) linearity MSM(c) == MSM(c_lo) + MSM(c_hi) and MSM(c) + MSM(r - c) == O on dense
        pseudo-random scalars (checks every bucket path at full size)."""
    n = 1 << log_n
    r = F.R_MOD
    B = gpu.Bases.srs_powers(fr_to_np([TAU])[0], n)
    # (1) sparse
    idx = [0, 1, 2, n // 3, n // 2 + 1, n - 2, n - 1]
    vals = rand_fr(len(idx), 3)
    dense = np.zeros((n, 4), dtype=np.uint64)
    dense[idx] = fr_to_np(vals)
    want = sum(v * pow(TAU, i, r) for i, v in zip(idx, vals)) % r
    assert jac_np_to_affine(gpu.msm(B, dense)) == EC.scalar_mul(EC.G1_GEN, want)
    # (2) dense canonical scalars < 2^253
    rng = np.random.default_rng(log_n)
    sc = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
    sc[:, 3] &= np.uint64((1 << 61) - 1)
    full = jac_np_to_affine(gpu.msm(B, sc, montgomery=False))
    lo = sc.copy(); lo[:, 2:] = 0
    hi = sc.copy(); hi[:, :2] = 0
    s = EC.add(jac_np_to_affine(gpu.msm(B, lo, montgomery=False)), jac_np_to_affine(gpu.msm(B, hi, montgomery=False)))
    assert full is not None and EC.is_on_curve(full) and full == s


@pytest.mark.parametrize("log_n", [13, 20, 22])
def test_fixed_base_commit_full_size(gpu, log_n):
    """the same known-tau and linearity properties through the fixed-base path (window table with the automatic width:
    12 / 19 / 20 bits), plus equality with the variable-base result on dense scalars and on a vector with a long zero
    gap (the shape of an opening proof's merged witness)."""
    n = 1 << log_n
    r = F.R_MOD
    B = gpu.Bases.srs_powers(fr_to_np([TAU])[0], n)
    rng = np.random.default_rng(100 + log_n)
    sc = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
    sc[:, 3] &= np.uint64((1 << 61) - 1)
    gap = sc.copy(); gap[n // 4: 3 * n // 4] = 0
    vb_full = jac_np_to_affine(gpu.msm(B, sc, montgomery=False))
    vb_gap = jac_np_to_affine(gpu.msm(B, gap, montgomery=False))
    B.precompute()
    assert B.table_info()[0] == min(20, log_n - 1)
    fb0, vb0 = gpu.msm_path_counts()
    assert jac_np_to_affine(gpu.msm(B, sc, montgomery=False)) == vb_full
    assert jac_np_to_affine(gpu.msm(B, gap, montgomery=False)) == vb_gap
    idx = [0, 1, 2, n // 3, n // 2 + 1, n - 2, n - 1]
    vals = rand_fr(len(idx), 5)
    # a sparse vector is too lightly loaded for the shared bucket set and takes the variable-base path; a dense one with
    # known structure exercises the table: p(X) = sum_i c X^i has p(tau) = c (tau^n - 1) / (tau - 1)
    cval = vals[0]
    const = np.tile(fr_to_np([cval]), (n, 1))
    want = cval * (pow(TAU, n, r) - 1) * pow(TAU - 1, -1, r) % r
    got = jac_np_to_affine(gpu.msm(B, const))
    assert got == EC.scalar_mul(EC.G1_GEN, want)
    # halves
    lo = sc.copy(); lo[n // 2:] = 0
    hi = sc.copy(); hi[:n // 2] = 0
    s = EC.add(jac_np_to_affine(gpu.msm(B, lo, montgomery=False)), jac_np_to_affine(gpu.msm(B, hi, montgomery=False)))
    assert s == vb_full
    fb1, vb1 = gpu.msm_path_counts()
    assert fb1 - fb0 >= 4          # dense, gap and the halves ran on the table (the constant vector is skewed -> fallback)
