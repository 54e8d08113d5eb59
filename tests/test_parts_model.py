"""Integer model of the virtual slots of the fixed-base accumulate kernel (marlin_amd/csrc/msm_fb.cuh: VTab, size_vscan_kernel,
vclass_of, accum30v_kernel, merge_parts_kernel): with the buckets ordered by size (largest first) and every bucket above Ts entries
cut into ceil(size class / T) parts, the map slot -> (bucket, part) -> entry range must cover every bucket's list exactly once, the
cut buckets must be perm's first `nsplit` with their parts in the first slots (slot = index into the parts' buffer), and the device's
rule for a buffer that is too small (double T and Ts) must end.  Mirrors the kernels statement by statement; the GPU tests check
the kernels themselves (MH_CHECK level 2 recomputes every bucket from its list)."""
import random

import pytest

SIZE_BINS = 1024


def vparts(s, T, Ts):
    return 1 if s <= Ts else (s + T - 1) // T


def vscan(sizes, T0, Ts0, vcap):
    """size_hist + size_vscan + size_perm: returns (perm, voff, pos, V, T, Ts, nsplit)"""
    cnt = [0] * SIZE_BINS
    for sz in sizes:
        cnt[min(sz, SIZE_BINS - 1)] += 1
    T = max(1, T0); Ts = max(T, Ts0)
    while sum(cnt[s] * vparts(s, T, Ts) for s in range(SIZE_BINS) if s > Ts) > vcap:
        T *= 2; Ts *= 2
    pos, voff = [0] * (SIZE_BINS + 1), [0] * (SIZE_BINS + 1)
    for i in range(SIZE_BINS):                       # index i = class SIZE_BINS - 1 - i: descending sizes
        s = SIZE_BINS - 1 - i
        pos[i + 1] = pos[i] + cnt[s]
        voff[i + 1] = voff[i] + cnt[s] * vparts(s, T, Ts)
    nsplit = pos[SIZE_BINS - 1 - min(Ts, SIZE_BINS - 1)] if Ts < SIZE_BINS - 1 else 0
    cur = list(pos)
    perm = [None] * len(sizes)
    for g, sz in enumerate(sizes):
        i = SIZE_BINS - 1 - min(sz, SIZE_BINS - 1)
        perm[cur[i]] = g; cur[i] += 1
    return perm, voff, pos, voff[SIZE_BINS], T, Ts, nsplit


def vclass_of(tab, x):
    lo, hi = 0, SIZE_BINS
    while hi - lo > 1:
        mid = (lo + hi) >> 1
        if tab[mid] <= x:
            lo = mid
        else:
            hi = mid
    return lo


@pytest.mark.parametrize("shape", ["uniform", "bimodal", "one_giant", "all_empty", "beyond_last_bin"])
@pytest.mark.parametrize("T0,Ts0,vcap", [(8, 8, 1 << 20), (25, 33, 1 << 20), (37, 49, 64), (300, 400, 1 << 20), (8, 12, 0)])
def test_virtual_slots_cover_every_list_once(shape, T0, Ts0, vcap):
    rnd = random.Random(hash((shape, T0, vcap)) & 0xffff)
    n = 700
    if shape == "uniform":
        sizes = [rnd.randrange(10, 40) for _ in range(n)]
    elif shape == "bimodal":
        sizes = [rnd.choice((18, 34, 54, 102)) + rnd.randrange(-3, 4) for _ in range(n)]
    elif shape == "one_giant":
        sizes = [rnd.randrange(0, 30) for _ in range(n)]; sizes[123] = 900
    elif shape == "all_empty":
        sizes = [0] * n
    else:
        sizes = [rnd.randrange(0, 50) for _ in range(n)]; sizes[5] = 4000; sizes[77] = 1023; sizes[78] = 1500
    perm, voff, pos, V, T, Ts, nsplit = vscan(sizes, T0, Ts0, vcap)
    assert sorted(perm) == list(range(n))
    assert all(min(sizes[perm[k]], SIZE_BINS - 1) >= min(sizes[perm[k + 1]], SIZE_BINS - 1) for k in range(n - 1))        # largest first
    cut_slots = sum(vparts(min(sz, SIZE_BINS - 1), T, Ts) for sz in sizes if min(sz, SIZE_BINS - 1) > Ts)
    assert cut_slots <= max(vcap, 0) or Ts >= SIZE_BINS - 1
    covered = [[] for _ in range(n)]
    parts_seen = {}
    for slot in range(V):                                # accum30v_kernel, one lane
        ci = vclass_of(voff, slot)
        assert voff[ci] <= slot < voff[ci + 1]
        p = vparts(SIZE_BINS - 1 - ci, T, Ts)
        rel = slot - voff[ci]; r, j = divmod(rel, p)
        gid = perm[pos[ci] + r]
        c = sizes[gid]
        lo, hi = c * j // p, c * (j + 1) // p
        covered[gid].append((lo, hi))
        parts_seen.setdefault(gid, []).append((j, slot, p))
        if p > 1:
            assert slot < cut_slots                      # a cut bucket's parts live in the first slots: slot = index into the parts' buffer
    for g in range(n):
        rs = sorted(covered[g])
        assert rs[0][0] == 0 and rs[-1][1] == sizes[g] and all(a[1] == b[0] for a, b in zip(rs, rs[1:])), (g, sizes[g], rs)
    for i in range(n):                                   # merge_parts_kernel: perm's first nsplit are exactly the cut buckets, parts adjacent
        g = perm[i]
        ps = sorted(parts_seen[g])
        assert (len(ps) > 1 or ps[0][2] > 1) == (i < nsplit), (i, nsplit, ps)
        if i < nsplit:
            ci = vclass_of(pos, i)
            p = vparts(SIZE_BINS - 1 - ci, T, Ts)
            first = voff[ci] + (i - pos[ci]) * p
            assert [x[1] for x in ps] == list(range(first, first + p)) and p == ps[0][2]
