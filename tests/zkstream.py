"""numpy restatement of the zk_rng stream `Marlin::prove` consumes (test infrastructure, oracle side).

The sequential generator lives in oracle/fs.py (ChaChaRng + fr_rand: rand_chacha 0.3 BlockRng word order and
ark-ff 0.3 `Fp256::rand` rejection sampling [SURVEY.md Appendix B-7]); it is pure Python and draws ~25 k field elements
per second, far too slow for the 3|H| mask-polynomial draws of a 2^18-constraint proof.  This module produces the SAME
stream vectorised over ChaCha blocks (checked against oracle/fs.py in tests/test_host_logic.py) so that the GPU tests
can pin the device's `DensePolynomial::rand` (rng.cuh: parallel blocks + rejection as stream compaction) and every
hiding polynomial of the commitments at full size.

Every draw of the zk_rng on the prove path is an Fr draw = 8 consecutive u32 words, so the
stream is: candidate i = words [8i, 8i + 8) of the ChaCha word stream (block counter from 0, stream id 0), top
REPR_SHAVE_BITS cleared, accepted iff < r; the accepted limbs are the Montgomery representation.
"""
import struct
import numpy as np
from oracle import fields as F

_U32 = np.uint32


def _rotl(x, n):
    return (x << _U32(n)) | (x >> _U32(32 - n))


def chacha_words(seed32, rounds, first_block, nblocks):
    """(nblocks, 16) uint32: ChaCha blocks first_block .. first_block + nblocks of the key `seed32`."""
    key = struct.unpack("<8I", bytes(seed32))
    ctr = np.arange(first_block, first_block + nblocks, dtype=np.uint64)
    st = [np.full(nblocks, c, dtype=_U32) for c in (0x61707865, 0x3320646e, 0x79622d32, 0x6b206574)]
    st += [np.full(nblocks, k, dtype=_U32) for k in key]
    st += [(ctr & np.uint64(0xFFFFFFFF)).astype(_U32), (ctr >> np.uint64(32)).astype(_U32),
           np.zeros(nblocks, dtype=_U32), np.zeros(nblocks, dtype=_U32)]
    w = [s.copy() for s in st]

    def qr(a, b, c, d):
        w[a] += w[b]; w[d] = _rotl(w[d] ^ w[a], 16)
        w[c] += w[d]; w[b] = _rotl(w[b] ^ w[c], 12)
        w[a] += w[b]; w[d] = _rotl(w[d] ^ w[a], 8)
        w[c] += w[d]; w[b] = _rotl(w[b] ^ w[c], 7)
    with np.errstate(over="ignore"):
        for _ in range(rounds // 2):
            qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15)
            qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14)
        out = np.stack([w[i] + st[i] for i in range(16)], axis=1)
    return out


def fr_draws(seed32, n, rounds=20):
    """The first n accepted `Fr::rand` draws of ChaChaRng::from_seed(seed32): (n, 4) uint64 limbs, which are the
    Montgomery representation of the drawn elements (ark-ff uses the accepted limbs as-is)."""
    r_limbs = [(F.R_MOD >> (64 * k)) & 0xFFFFFFFFFFFFFFFF for k in range(4)]
    top_mask = np.uint64((1 << (64 - F.FR_REPR_SHAVE_BITS)) - 1)
    got, have, blk = [], 0, 0
    while have < n:
        ncand = max(1024, int((n - have) * 1.15) + 64)
        ncand += ncand & 1                                   # two candidates per 16-word block
        words = chacha_words(seed32, rounds, blk, ncand // 2).reshape(-1, 8).astype(np.uint64)
        blk += ncand // 2
        limbs = words[:, 0::2] | (words[:, 1::2] << np.uint64(32))
        limbs[:, 3] &= top_mask
        lt = np.zeros(len(limbs), dtype=bool)
        eq = np.ones(len(limbs), dtype=bool)
        for k in (3, 2, 1, 0):
            rk = np.uint64(r_limbs[k])
            lt |= eq & (limbs[:, k] < rk)
            eq &= limbs[:, k] == rk
        acc = limbs[lt]
        got.append(acc)
        have += len(acc)
    return np.concatenate(got)[:n]


def limbs_to_int(row):
    return int(row[0]) | (int(row[1]) << 64) | (int(row[2]) << 128) | (int(row[3]) << 192)


def mont_to_canonical(row):
    return F.fr_from_mont(limbs_to_int(row))


def prove_zk_draws(seed32, H, rounds=20):
    """The zk_rng draws of one MarlinKZG10 prove with |domain_h| = H, in the order of SURVEY.md Appendix C:
    returns dict(r_w, r_za, r_zb: canonical ints; mask: (3H, 4) uint64 Montgomery coefficients AFTER the
    sum-over-H fix of prover.rs:373-380; blind_w, blind_za, blind_zb, blind_g1, blind_g1_shifted: 3 canonical ints each)."""
    d = fr_draws(seed32, 3 + 3 * H + 15, rounds)
    out = {"r_w": mont_to_canonical(d[0]), "r_za": mont_to_canonical(d[1]), "r_zb": mont_to_canonical(d[2])}
    mask = d[3:3 + 3 * H].copy()
    # mask_poly[0] -= sum_{i = 0..upper_bound} mask_poly[nh * i], upper_bound = (3H - 1) / H = 2
    r0 = sum(mont_to_canonical(mask[H * i]) for i in range(3)) % F.R_MOD
    m0 = F.fr_to_mont((mont_to_canonical(mask[0]) - r0) % F.R_MOD)
    mask[0] = [(m0 >> (64 * k)) & 0xFFFFFFFFFFFFFFFF for k in range(4)]
    out["mask"] = mask
    o = 3 + 3 * H
    for name in ("blind_w", "blind_za", "blind_zb", "blind_g1", "blind_g1_shifted"):
        out[name] = [mont_to_canonical(d[o + k]) for k in range(3)]
        o += 3
    return out
