"""Integer model of the fixed-base bucket reduction of marlin_amd/csrc/msm_fb.cuh (rsum_kernel, plane_kernel) and of the plan /
coefficients FbRun::prepare builds for it (capi.hip): with integers in place of bucket points, row / column sums by lane groups,
bit planes of the row and column indices and the host's coefficients must reproduce sum_b (b + 1) S_b over the OWNED buckets --
every window width, one GPU and bucket-range shards of 2 / 3 / 4 / 8 ranks, thin and full launches.  This is a model of the index
arithmetic (the GPU tests check the kernels themselves against the oracle); it mirrors the kernels statement by statement."""
import random

import pytest


def insert_one(x, p):
    return ((x >> p) << (p + 1)) | (1 << p) | (x & ((1 << p) - 1))


def plan(c, first, stride, nj, num_simds):
    nbt = 1 << (c - 1)
    pshift = min(c - 1, 11)
    nb = 1 << pshift
    nparts = nbt // nb
    if stride > 1 and nparts < stride:
        first, stride = 0, 1
    nown = (nparts - first + stride - 1) // stride
    nbown = nown * nb
    lgown = 0
    while (1 << lgown) < nbown:
        lgown += 1
    lgC = max(1, min(pshift, (lgown + 1) // 2))
    C = 1 << lgC
    lgrpp = pshift - lgC
    R = nbown >> lgC
    lgM = 0
    while (1 << lgM) < R:
        lgM += 1
    per_simd = 2 * nj * nbown
    Lt = (per_simd + 128 * num_simds - 1) // (128 * num_simds)
    if Lt < 16:
        Lt = min(16, (per_simd + 64 * num_simds - 1) // (64 * num_simds))
    Lt = max(1, Lt)

    def lanes(length):
        lg = 0
        while lg < 6 and (2 << lg) * Lt <= length:
            lg += 1
        return lg
    lgJ = lanes(C); J = 1 << lgJ; Lr = C >> lgJ
    lgI = lanes(1 << lgM); I = 1 << lgI; Lc = (R + I - 1) // I
    NTr = (R * J + 63) & ~63
    NT = NTr + ((C * I + 63) & ~63)
    npl = lgC + lgM + 1
    coef = [0] * npl
    for p in range(lgC):
        coef[p] = 1 << p
    for p in range(lgM):
        coef[lgC + p] = (C << p) * (1 if p < lgrpp else stride)
    coef[npl - 1] = ((C * first) << lgrpp) + 1
    return dict(nbt=nbt, nb=nb, nparts=nparts, first=first, stride=stride, lgC=lgC, C=C, lgrpp=lgrpp, R=R, lgM=lgM, lgJ=lgJ, J=J, Lr=Lr,
                lgI=lgI, I=I, Lc=Lc, NTr=NTr, NT=NT, npl=npl, coef=coef)


def reduce_model(P, S):
    C, R, first, stride, lgrpp, lgC = P["C"], P["R"], P["first"], P["stride"], P["lgrpp"], P["lgC"]
    sums = [0] * (R + C)
    for wave in range(P["NT"] // 64):
        acc = [0] * 64
        lane = []
        for l in range(64):
            q = wave * 64 + l
            if q < P["NTr"]:
                grp, g, lgG, L = q >> P["lgJ"], q & (P["J"] - 1), P["lgJ"], P["Lr"]
                lane.append([True, grp, g, grp, g, 0, P["J"], grp < R])
            else:
                q2 = q - P["NTr"]
                grp, g, lgG, L = q2 >> P["lgI"], q2 & (P["I"] - 1), P["lgI"], P["Lc"]
                lane.append([False, grp, g, g, grp, P["I"], 0, grp < C])
        for step in range(L + lgG):                       # L, lgG of the last lane: the wave is homogeneous
            if step < L:
                for l, st in enumerate(lane):
                    row, grp, g, m, c, dm, dc, valid = st
                    if valid and m < R and c < C:
                        v = first + (m >> lgrpp) * stride
                        r = (v << lgrpp) | (m & ((1 << lgrpp) - 1))
                        acc[l] += S[(r << lgC) | c]
                    st[3] += dm; st[4] += dc
            else:
                mask = 1 << (step - L)
                acc = [acc[l] + acc[l ^ mask] for l in range(64)]
        for l, st in enumerate(lane):
            if st[7] and st[2] == 0:
                sums[st[1] if st[0] else R + st[1]] = acc[l]
    planes = []
    for pl in range(P["npl"]):
        if pl < lgC:
            qs = [R + insert_one(k, pl) for k in range(C >> 1)]
        elif pl < lgC + P["lgM"]:
            qs = [q for q in (insert_one(k, pl - lgC) for k in range(1 << (P["lgM"] - 1))) if q < R]
        else:
            qs = list(range(R))
        planes.append(sum(sums[q] for q in qs))
    return sum(cf * x for cf, x in zip(P["coef"], planes))


@pytest.mark.parametrize("c", [4, 5, 8, 12, 13, 16, 17])
@pytest.mark.parametrize("shard", [(0, 1), (1, 2), (2, 3), (3, 4), (5, 8)])
@pytest.mark.parametrize("nj,num_simds", [(1, 1024), (4, 1024), (4, 8), (8, 64)])
def test_row_column_sums_and_bit_planes_reproduce_the_weighted_bucket_sum(c, shard, nj, num_simds):
    P = plan(c, shard[0], shard[1], nj, num_simds)
    rnd = random.Random(c * 1000 + shard[0] * 10 + nj)
    S = [rnd.randrange(1 << 24) for _ in range(P["nbt"])]
    want = sum((b + 1) * S[b] for v in range(P["first"], P["nparts"], P["stride"]) for b in range(v * P["nb"], (v + 1) * P["nb"]))
    assert P["J"] * P["Lr"] == P["C"] and P["I"] * P["Lc"] >= P["R"] and P["NTr"] % 64 == 0 and P["NT"] % 64 == 0
    assert reduce_model(P, S) == want
