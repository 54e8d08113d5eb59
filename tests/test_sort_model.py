"""CPU model of the one-pass partition's tile sizing (marlin_amd/csrc/capi.hip: FbRun::prepare, round 6).

The sort of the fixed-base MSM no longer counts before it splits: every split block writes its entries, grouped by virtual
window, into its own region, and a hist / scatter tile is the window's runs in `bpt` consecutive split blocks, `bpt` chosen on the
HOST from the load a uniformly distributed scalar puts on the window (nothing is fetched from the device).  This file restates
that rule and the signed-digit recoding of msm.cuh on integers (numpy) and checks, for the window widths the prover uses:

* the model's expected entries per (split block, window) match what recoding random scalars mod r gives, window by window;
* with the 90 % fill the rule aims at, every tile of every window stays inside the 16 384-entry staging area -- for scalars
  uniform mod r (what a prover's coefficient vectors are) with a margin of many standard deviations;
* scalars that are NOT uniform mod r (below 2^251: the top window's digits all fall into the lowest partitions) overflow tiles by
  a wide margin -- which is why the scatter works an overfull tile off in sub-tiles instead of trusting the model.

The function these lists feed replaces VariableBaseMSM::multi_scalar_mul (/root/reference src/lib.rs:172,193,213,292)."""
import numpy as np
import pytest

R_MOD = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
MAX_TILE, SPLIT_ENTRIES, PART_BITS, MAX_BPT, FILL = 16 * 1024, 13312, 11, 4096, 0.90


def make_windows(c):
    W = (256 + c - 1) // c
    narrow = W * c - 256
    start, bits, bit = [], [], 0
    for w in range(W):
        wb = c - 1 if w >= W - narrow else c
        start.append(bit); bits.append(wb); bit += wb
    return W, start, bits


def split_scalars(W):
    return min(1024, (SPLIT_ENTRIES // W) & ~63)


def plan(c):
    """(W, start, bits, S, nb, nparts, bpt[v], per_block[v]) exactly as FbRun::prepare computes them"""
    W, start, bits = make_windows(c)
    S = split_scalars(W)
    pshift = min(c - 1, PART_BITS)
    nb = 1 << pshift
    nparts = (1 << (c - 1)) // nb
    per_block, bpt = [], []
    for v in range(nparts):
        e = 0.0
        for w in range(W):
            top = float(1 << (bits[w] - 1))
            lo, hi = float(v * nb), min(float((v + 1) * nb), top)
            if hi > lo:
                e += S * (hi - lo) * 2.0 / float(1 << bits[w])
        per_block.append(e)
        bpt.append(int(max(1.0, min(FILL * MAX_TILE / max(e, 1e-9), float(MAX_BPT)))))
    return W, start, bits, S, nb, nparts, bpt, per_block


def recode(limbs, W, start, bits):
    """msm::for_each_digit on (n, 4) uint64 limbs: (n, W) arrays of |digit| (0 = no entry)"""
    n = limbs.shape[0]
    words = limbs.view(np.uint32).reshape(n, 8).astype(np.uint64)
    carry = np.zeros(n, dtype=np.uint64)
    out = np.zeros((n, W), dtype=np.uint32)
    for w in range(W):
        bit, wb = start[w], bits[w]
        limb, sh = bit >> 5, bit & 31
        two = words[:, limb].copy()
        if limb + 1 < 8:
            two |= words[:, limb + 1] << np.uint64(32)
        raw = ((two >> np.uint64(sh)) & np.uint64((1 << wb) - 1)) + carry
        neg = raw > np.uint64(1 << (wb - 1))
        out[:, w] = np.where(neg, np.uint64(1 << wb) - raw, raw).astype(np.uint32)
        carry = neg.astype(np.uint64)
    return out


def uniform_mod_r(rng, n):
    x = rng.integers(0, 1 << 63, size=(2 * n + 64, 4), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(2 * n + 64, 4), dtype=np.uint64)
    top = (R_MOD >> 192)
    x[:, 3] &= np.uint64((1 << 63) - 1)                       # 255 bits
    keep = x[:, 3] < np.uint64(top)                           # (the boundary limb itself: 2^-63 of the mass, ignored)
    return np.ascontiguousarray(x[keep][:n])


def tiles_of(digits, S, nb, nparts, bpt):
    """entries of every (window v, tile) for the given |digit| matrix: dict v -> list of tile sizes"""
    n = digits.shape[0]
    nblk = (n + S - 1) // S
    blk = np.arange(n) // S
    out = {}
    part = (digits.astype(np.int64) - 1) // nb                # -1 for zero digits
    for v in range(nparts):
        per_blk = np.bincount(blk.repeat(digits.shape[1])[(part == v).ravel()], minlength=nblk)
        out[v] = [int(per_blk[t:t + bpt[v]].sum()) for t in range(0, nblk, bpt[v])]
    return out


@pytest.mark.parametrize("c", [20, 16])
def test_tile_rule_matches_recoded_uniform_scalars(c):
    W, start, bits, S, nb, nparts, bpt, per_block = plan(c)
    assert S * W <= SPLIT_ENTRIES <= MAX_TILE
    rng = np.random.default_rng(2026 + c)
    n = 400 * S if c == 20 else 64 * S                        # c = 20: more than two tiles of every window
    digits = recode(uniform_mod_r(rng, n), W, start, bits)
    nblk = n // S
    part = (digits.astype(np.int64) - 1) // nb
    worst = 0.0
    for v in range(nparts):
        got = float((part == v).sum()) / nblk
        # the model ignores that the top window's scalars stop at r = 0.906 * 2^255, not at 2^255: its digits are 10 % denser on the
        # partitions below r's top digits and absent above them -- a little more / up to S * 2^11 * 2 / 2^19 = 8 fewer entries per
        # block than planned, both inside the margin
        assert per_block[v] * 0.85 - 1.5 <= got <= 1.04 * per_block[v] + 1.5, (c, v, got, per_block[v])
        worst = max(worst, got / max(per_block[v], 1e-9))
    assert worst < 1.05
    tiles = tiles_of(digits, S, nb, nparts, bpt)
    biggest = max(max(t) for t in tiles.values())
    assert biggest <= MAX_TILE, biggest
    if c == 20:
        full = [t[0] for v, t in tiles.items() if len(t) > 1]             # tiles of bpt whole blocks
        assert full and min(full) > 0.75 * MAX_TILE and max(full) < 0.97 * MAX_TILE, (min(full), max(full))   # (the lightest: partitions above the top digits of r)


def test_scalars_below_2p251_overflow_the_model_and_need_the_sub_tiles():
    """what tests/test_gpu_msm.py's 2^22-point case feeds the sort: uniform 251-bit values.  The top window (bits 237 ... 255) then
    holds 14-bit digits only -- every one of them in the lowest eight partitions, eight times the load the rule plans for."""
    c = 20
    W, start, bits, S, nb, nparts, bpt, per_block = plan(c)
    rng = np.random.default_rng(5)
    n = 2 * bpt[0] * S
    x = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
    x[:, 3] &= np.uint64((1 << 59) - 1)
    tiles = tiles_of(recode(x, W, start, bits), S, nb, nparts, bpt)
    assert max(tiles[0]) > 2 * MAX_TILE                                     # the scatter's sub-tiles take it from here
    assert max(tiles[nparts - 1]) <= MAX_TILE
