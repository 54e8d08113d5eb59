"""Device SRS generation ([tau^i]G) and the known-tau KZG identity commit(p) == [p(tau)]G
(SURVEY.md §8c item 2): an O(1) check of an MSM of any size."""
import numpy as np
import pytest
from oracle import fields as F, curve as EC, poly as OP
from tests.util import fr_to_np, jac_np_to_affine, rand_fr, limbs_to_fq, points_to_np

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
    13 / 19 / 20 bits), plus equality with the variable-base result on dense scalars and on a vector with a long zero
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
    from tests.util import auto_window_bits
    assert B.table_info()[0] == auto_window_bits(n)
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


@pytest.mark.parametrize("compressed", [True, False])
def test_serialized_srs_decodes_on_the_device(gpu, compressed):
    """`Vec<G1Affine>` in ark-serialize's format (powers_of_g of a serialized SRS) uploaded as bytes and decoded by the
    device (square root and sign selection per point) equals the points themselves; every way an item can be invalid
    fails the call (SerializationError::InvalidData)."""
    from oracle import marlin as MR, fields as F
    from marlin_amd.api import Bases
    import marlin_amd as M
    n = 300
    pts = EC.srs_powers(0x1f3a9c5d7e2b4a6f, n)                 # both signs of y occur
    fq = F.FQ_BYTES

    def image(p):
        if compressed:
            return MR.g1_compressed(p)
        return p[0].to_bytes(fq, "little") + p[1].to_bytes(fq, "little")
    data = b"".join(image(p) for p in pts)
    if compressed:
        assert len({b[-1] >> 7 for b in (image(p) for p in pts)}) == 2
    b = Bases.from_serialized(data, n, compressed)
    assert (b.download() == points_to_np(pts)).all()
    # and it multiplies like the same points uploaded as limbs
    sc = rand_fr(n, 3)
    d = M.DeviceBuffer.from_numpy(fr_to_np(sc))
    assert jac_np_to_affine(M.msm_dev(b, d, n)) == EC.msm_naive(pts, sc)
    b.free()
    item = len(data) // n

    def corrupt(k, f):
        t = bytearray(data)
        t[k * item:(k + 1) * item] = f(bytearray(t[k * item:(k + 1) * item]))
        return bytes(t)
    bad = []
    bad.append(corrupt(7, lambda e: bytearray((F.Q_MOD).to_bytes(fq, "little")) + e[fq:]))          # x = p: not canonical
    if compressed:
        bad.append(corrupt(11, lambda e: bytearray(b"\x00" * (fq - 1) + b"\x40")))                   # the identity
        x = 1
        while pow((x ** 3 + F.G1_B) % F.Q_MOD, (F.Q_MOD - 1) // 2, F.Q_MOD) == 1:
            x += 1
        bad.append(corrupt(13, lambda e: bytearray(x.to_bytes(fq, "little"))))                       # x^3 + b is not a square
        bad.append(corrupt(17, lambda e: e[:-1] + bytes([e[-1] | 0xC0])))                            # both flags
    else:
        bad.append(corrupt(19, lambda e: e[:fq] + bytearray(((int.from_bytes(e[fq:], "little") + 1) % F.Q_MOD).to_bytes(fq, "little"))))   # off the curve
        bad.append(corrupt(23, lambda e: e[:-1] + bytes([e[-1] | 0x40])))                            # infinity flag on y
    for t in bad:
        with pytest.raises(Exception, match="do not decode|InvalidData"):
            Bases.from_serialized(t, n, compressed)
    # a flipped sign bit decodes to the negated point, not to an error
    if compressed:
        t = corrupt(5, lambda e: e[:-1] + bytes([e[-1] ^ 0x80]))
        b = Bases.from_serialized(t, n, True)
        got = b.download()
        want = points_to_np(pts[:5] + [EC.neg(pts[5])] + pts[6:])
        assert (got == want).all()
        b.free()
