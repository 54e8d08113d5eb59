"""Integer model of the fixed-base bucket set and its reduction (marlin_amd/csrc/msm_fb.cuh: digit_pos, rsum_kernel, plane_kernel)
and of the plan / coefficients FbRun::prepare builds for them (capi.hip).  With integers in place of bucket points:
* digit_pos maps every digit value |d| in [1, 2^(c-1)] to a (position, doubled) pair with weight(position) x multiplier = d, onto
  exactly 3 * 2^(c-3) positions (level A: weights 2k + 1, the digits = 2 (mod 4) arriving halved against the doubled table point;
  level B: weights 4 (k + 1));
* row / column sums by lane groups (xor butterflies), the bit planes of the row and column indices of both levels (column planes of
  equal coefficient merged) and the host's coefficients reproduce sum_pos weight(pos) S_pos over the OWNED buckets -- every
  window width, one GPU and bucket-range shards of 2 / 3 / 4 / 8 ranks, thin and full launches.
This is a model of the index arithmetic (the GPU tests check the kernels themselves against the oracle); it mirrors the kernels and the
plan statement by statement."""
import random

import pytest


def insert_one(x, p): return ((x >> p) << (p + 1)) | (1 << p) | (x & ((1 << p) - 1))
def digit_pos(b, lgA):
    if b & 1: return b >> 1, 0
    if b & 2: return b >> 2, 1
    return (1 << lgA) + (b >> 2) - 1, 0
def weight(pos, lgA):
    return 2 * pos + 1 if pos < (1 << lgA) else 4 * (pos - (1 << lgA) + 1)
def plan(c, first, stride, nj, num_simds):
    lgA = c - 2
    nbt = 3 << (c - 3)
    pshift = min(c - 3, 11)
    nb = 1 << pshift
    nparts = nbt // nb
    if stride > 1 and nparts < stride: first, stride = 0, 1
    npA = (1 << lgA) // nb
    lv = []
    for (p0, npl) in ((0, npA), (npA, nparts - npA)):
        fv = p0 + ((first - p0) % stride)
        n = (p0 + npl - fv + stride - 1) // stride if fv < p0 + npl else 0
        lv.append(dict(p0=p0, npl=npl, fv=fv, nown=n))
    nbown = sum(l["nown"] for l in lv) * nb
    lgown = 0
    while (1 << lgown) < max(nbown, 1): lgown += 1
    lgC = max(1, min(pshift, (lgown + 1) // 2))
    C = 1 << lgC; lgrpp = pshift - lgC; rpp = 1 << lgrpp
    per_simd = 2 * nj * nbown
    Lt = (per_simd + 128 * num_simds - 1) // (128 * num_simds)
    if Lt < 16: Lt = min(16, (per_simd + 64 * num_simds - 1) // (64 * num_simds))
    Lt = max(1, Lt)
    def lanes(length):
        lg = 0
        while lg < 6 and (2 << lg) * Lt <= length: lg += 1
        return lg
    lgJ = lanes(C); J = 1 << lgJ; Lr = C >> lgJ
    soff = 0; toff = 0
    for l in lv:
        R = l["nown"] * rpp
        lgM = 0
        while (1 << lgM) < R: lgM += 1
        lgI = lanes(1 << lgM) if R else 0; I = 1 << lgI; Lc = (R + I - 1) // I if R else 0
        NTr = (R * J + 63) & ~63
        NT = NTr + ((C * I + 63) & ~63) if R else 0
        if not R: NTr = 0
        l.update(R=R, lgM=lgM, lgI=lgI, I=I, Lc=Lc, NTr=NTr, NT=NT, soff=soff, toff=toff)
        soff += R + (C if R else 0); toff += NT
    P = dict(lgA=lgA, nbt=nbt, nb=nb, pshift=pshift, nparts=nparts, first=first, stride=stride, lgC=lgC, C=C, lgrpp=lgrpp, lgJ=lgJ, J=J, Lr=Lr, lv=lv, NS=soff, NT=toff)
    # planes: col g = 1..lgC+1, rowA bits, rowB bits, TA, TB
    planes = [("col", g) for g in range(1, lgC + 2)] + [("row", 0, p) for p in range(lv[0]["lgM"])] + [("row", 1, p) for p in range(lv[1]["lgM"])] + [("tot", 0), ("tot", 1)]
    coef = []
    for pl in planes:
        if pl[0] == "col": coef.append(1 << pl[1])
        elif pl[0] == "row":
            lvl, p = pl[1], pl[2]
            coef.append((2 << lvl) * (C << p) * (1 if p < lgrpp else stride))
        else:
            lvl = pl[1]; l = lv[lvl]
            a = (l["fv"] - l["p0"]) * rpp
            coef.append((1 << (2 * lvl)) + (2 << lvl) * C * a)
    P["planes"] = planes; P["coef"] = coef
    return P
def reduce_model(P, S):
    C, lgC, lgrpp, stride = P["C"], P["lgC"], P["lgrpp"], P["stride"]
    sums = [0] * P["NS"]
    for wave in range(P["NT"] // 64):
        acc = [0] * 64; lane = []
        for ln in range(64):
            q = wave * 64 + ln
            lvl = 0 if q < P["lv"][0]["NT"] else 1
            l = P["lv"][lvl]; q -= l["toff"]
            if q < l["NTr"]:
                grp, g, lgG, L = q >> P["lgJ"], q & (P["J"] - 1), P["lgJ"], P["Lr"]
                lane.append([True, grp, g, grp, g, 0, P["J"], grp < l["R"], l])
            else:
                q2 = q - l["NTr"]
                grp, g, lgG, L = q2 >> l["lgI"], q2 & (l["I"] - 1), l["lgI"], l["Lc"]
                lane.append([False, grp, g, g, grp, l["I"], 0, grp < C, l])
        for step in range(L + lgG):
            if step < L:
                for ln, st in enumerate(lane):
                    row, grp, g, m, c, dm, dc, valid, l = st
                    if valid and m < l["R"] and c < C:
                        v = l["fv"] + (m >> lgrpp) * stride
                        r = (v << lgrpp) | (m & ((1 << lgrpp) - 1))
                        acc[ln] += S[(r << lgC) | c]
                    st[3] += dm; st[4] += dc
            else:
                mask = 1 << (step - L)
                acc = [acc[ln] + acc[ln ^ mask] for ln in range(64)]
        for ln, st in enumerate(lane):
            if st[7] and st[2] == 0:
                l = st[8]
                sums[l["soff"] + (st[1] if st[0] else l["R"] + st[1])] = acc[ln]
    out = []
    for pl in P["planes"]:
        qs = []
        if pl[0] == "col":
            g = pl[1]
            for lvl, p in ((0, g - 1), (1, g - 2)):
                l = P["lv"][lvl]
                if l["R"] and 0 <= p < lgC:
                    qs += [l["soff"] + l["R"] + insert_one(k, p) for k in range(C >> 1)]
        elif pl[0] == "row":
            l = P["lv"][pl[1]]; p = pl[2]
            qs = [l["soff"] + q for q in (insert_one(k, p) for k in range(1 << (l["lgM"] - 1))) if q < l["R"]]
        else:
            l = P["lv"][pl[1]]
            qs = [l["soff"] + m for m in range(l["R"])]
        out.append(sum(sums[q] for q in qs))
    return sum(cf * x for cf, x in zip(P["coef"], out))


@pytest.mark.parametrize("c", [4, 5, 6, 8, 12, 13, 14, 15, 16, 17])
@pytest.mark.parametrize("shard", [(0, 1), (1, 2), (2, 3), (3, 4), (5, 8)])
@pytest.mark.parametrize("nj,num_simds", [(1, 1024), (4, 1024), (4, 8), (8, 64)])
def test_row_column_sums_and_bit_planes_reproduce_the_weighted_bucket_sum(c, shard, nj, num_simds):
    P = plan(c, shard[0], shard[1], nj, num_simds)
    rnd = random.Random(c * 1000 + shard[0] * 10 + nj)
    S = [rnd.randrange(1 << 24) for _ in range(P["nbt"])]
    want = sum(weight(b, P["lgA"]) * S[b] for v in range(P["first"], P["nparts"], P["stride"]) for b in range(v * P["nb"], (v + 1) * P["nb"]))
    assert P["J"] * P["Lr"] == P["C"] and P["NT"] % 64 == 0 and all(l["NTr"] % 64 == 0 and l["I"] * l["Lc"] >= l["R"] for l in P["lv"])
    assert reduce_model(P, S) == want


@pytest.mark.parametrize("c", [4, 7, 13, 20])
def test_every_digit_value_has_one_bucket_and_a_multiplier(c):
    lgA = c - 2
    seen = set()
    for d in range(1, (1 << (c - 1)) + 1):
        pos, dbl = digit_pos(d, lgA)
        assert pos < (3 << (c - 3)) and weight(pos, lgA) * (2 if dbl else 1) == d, (c, d, pos, dbl)
        seen.add(pos)
    assert len(seen) == 3 << (c - 3)          # every position is used: no bucket with a weight = 2 (mod 4) exists
