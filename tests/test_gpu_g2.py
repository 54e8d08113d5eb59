"""G2 through the C ABI (mh_g2_*): multi-scalar multiplication and the fixed-base powers of KZG10::setup's G2 side
against the oracle's affine arithmetic (oracle/g2.py) -- naive sums at small n, the known-dlog identity at large n
(bases [tau^i]H  =>  sum s_i [tau^i]H = [sum s_i tau^i]H, O(1) oracle work), edge cases.  Replaces ark_ec
VariableBaseMSM over G2Affine / kzg10::setup(produce_g2_powers) (reached from /root/reference src/lib.rs:79-96)."""
import random
import numpy as np
import pytest
from oracle import fields as F, g2 as G2
from tests.util import fr_to_np, fq_to_limbs, limbs_to_fq, rand_fr

pytestmark = pytest.mark.gpu
L = F.FQ_LIMBS64
TAU = 0x1f3a9c5d7e2b4a6f8091a2b3c4d5e6f708192a3b


def g2_to_np(points):
    out = np.zeros((len(points), 4 * L), dtype=np.uint64)
    for i, ((x0, x1), (y0, y1)) in enumerate(points):
        out[i] = fq_to_limbs(x0) + fq_to_limbs(x1) + fq_to_limbs(y0) + fq_to_limbs(y1)
    return out


def np_to_g2(row, inf=False):
    if inf:
        return None
    return ((limbs_to_fq(row[0:L]), limbs_to_fq(row[L:2 * L])), (limbs_to_fq(row[2 * L:3 * L]), limbs_to_fq(row[3 * L:4 * L])))


def test_g2_srs_powers_match_oracle(gpu):
    """[scale tau^(first+i)]H, incl. a negative scale (SonicKZG10's neg_powers_of_h) and a `first` offset."""
    gen = g2_to_np([G2.G2_GEN])[0]
    B = gpu.G2Bases.srs_powers(gen, fr_to_np([TAU])[0], 6)
    got = [np_to_g2(r) for r in B.download()]
    assert got == [G2.g2_mul(G2.G2_GEN, pow(TAU, i, F.R_MOD)) for i in range(6)]
    assert all(G2.g2_is_on_curve(p) for p in got)
    neg = gpu.G2Bases.srs_powers(gen, fr_to_np([TAU])[0], 3, scale_mont=fr_to_np([F.R_MOD - 1])[0], first=100)
    assert [np_to_g2(r) for r in neg.download()] == [G2.g2_neg(G2.g2_mul(G2.G2_GEN, pow(TAU, 100 + i, F.R_MOD))) for i in range(3)]
    with pytest.raises(gpu.MarlinHipError, match="identity"):
        gpu.G2Bases.srs_powers(gen, fr_to_np([0])[0], 2, first=1)


@pytest.mark.parametrize("n", [1, 2, 33, 200])
def test_g2_msm_matches_naive(gpu, n):
    rng = random.Random(n)
    pts = [G2.g2_mul(G2.G2_GEN, rng.randrange(1, F.R_MOD)) for _ in range(min(n, 8))]
    pts = [pts[i % len(pts)] for i in range(n)]                      # repeated bases on purpose
    sc = [rng.randrange(F.R_MOD) for _ in range(n)]
    B = gpu.G2Bases(g2_to_np(pts))
    out, inf = gpu.g2_msm(B, fr_to_np(sc))
    assert np_to_g2(out, inf) == G2.g2_msm_naive(pts, sc)
    out2, inf2 = gpu.g2_msm(B, fr_to_np(sc, montgomery=False), montgomery=False)     # canonical scalars: same point
    assert np_to_g2(out2, inf2) == np_to_g2(out, inf)


@pytest.mark.parametrize("log_n", [10, 14, 16])
def test_g2_msm_known_dlog(gpu, log_n):
    n = 1 << log_n
    B = gpu.G2Bases.srs_powers(g2_to_np([G2.G2_GEN])[0], fr_to_np([TAU])[0], n)
    sc = rand_fr(n, 77 + log_n)
    want_k, t = 0, 1
    for s in sc:
        want_k = (want_k + s * t) % F.R_MOD
        t = t * TAU % F.R_MOD
    out, inf = gpu.g2_msm(B, fr_to_np(sc))
    assert np_to_g2(out, inf) == G2.g2_mul(G2.G2_GEN, want_k)
    # a slice of the base set (offset), as kzg10 slices powers
    off = 37
    sub = sc[: n - off]
    k2 = sum(s * pow(TAU, off + i, F.R_MOD) for i, s in enumerate(sub[:500])) % F.R_MOD
    out, inf = gpu.g2_msm(B, fr_to_np(sub[:500]), base_offset=off)
    assert np_to_g2(out, inf) == G2.g2_mul(G2.G2_GEN, k2)


def test_g2_msm_edge_cases(gpu):
    pts = [G2.g2_mul(G2.G2_GEN, k) for k in (3, 5)]
    B = gpu.G2Bases(g2_to_np([pts[0], pts[0], pts[1]]))
    for sc, want in [([0, 0, 0], None),                                         # all-zero scalars: the identity
                     ([7, F.R_MOD - 7, 0], None),                               # s P - s P
                     ([1, 1, 1], G2.g2_mul(G2.G2_GEN, 11)),
                     ([F.R_MOD - 1, 0, F.R_MOD - 1], G2.g2_neg(G2.g2_mul(G2.G2_GEN, 8))),
                     ([1 << 254 if F.R_MOD > (1 << 254) else (1 << 253), 2, 3], None)]:
        out, inf = gpu.g2_msm(B, fr_to_np(sc))
        if want is None and sc[0] >= (1 << 253):
            want = G2.g2_msm_naive([pts[0], pts[0], pts[1]], sc)
        assert np_to_g2(out, inf) == want
    out, inf = gpu.g2_msm(B, np.zeros((0, 4), dtype=np.uint64))                  # empty input
    assert inf
    bad = g2_to_np([pts[0]])
    bad[0, 0] ^= 1
    with pytest.raises(gpu.MarlinHipError, match="twist"):
        gpu.G2Bases(bad)
    with pytest.raises(gpu.MarlinHipError, match="twist"):
        gpu.G2Bases(np.zeros((1, 4 * L), dtype=np.uint64))                      # (0, 0): the identity cannot be a base
