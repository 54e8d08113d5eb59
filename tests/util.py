"""Test helpers: Python ints (oracle side) <-> numpy limb arrays (C-ABI side)."""
import random
import numpy as np
from oracle import fields as F
from oracle import curve as EC

MASK64 = (1 << 64) - 1


def fr_to_np(vals, montgomery=True):
    """list of canonical ints mod r -> (n,4) uint64 (Montgomery form by default)."""
    out = np.zeros((len(vals), 4), dtype=np.uint64)
    for i, v in enumerate(vals):
        x = F.fr_to_mont(v % F.R_MOD) if montgomery else v % F.R_MOD
        for k in range(4):
            out[i, k] = (x >> (64 * k)) & MASK64
    return out


def np_to_fr(arr, montgomery=True):
    arr = np.asarray(arr, dtype=np.uint64).reshape(-1, 4)
    out = []
    for row in arr:
        x = 0
        for k in range(4):
            x |= int(row[k]) << (64 * k)
        out.append(F.fr_from_mont(x) if montgomery else x)
    return out


def fq_to_limbs(x):
    x = F.fq_to_mont(x % F.Q_MOD)
    return [(x >> (64 * k)) & MASK64 for k in range(F.FQ_LIMBS64)]


def limbs_to_fq(limbs):
    x = 0
    for k in range(F.FQ_LIMBS64):
        x |= int(limbs[k]) << (64 * k)
    return F.fq_from_mont(x)


def points_to_np(points):
    """list of affine (x, y) ints -> (n,12) uint64 x||y Montgomery."""
    L = F.FQ_LIMBS64
    out = np.zeros((len(points), 2 * L), dtype=np.uint64)
    for i, (x, y) in enumerate(points):
        out[i, :L] = fq_to_limbs(x)
        out[i, L:] = fq_to_limbs(y)
    return out


def jac_np_to_affine(xyz):
    """(18,) uint64 Jacobian Montgomery -> oracle affine point (or None)."""
    L = F.FQ_LIMBS64
    X = limbs_to_fq(xyz[0:L])
    Y = limbs_to_fq(xyz[L:2 * L])
    Z = limbs_to_fq(xyz[2 * L:3 * L])
    return EC.jac_to_affine((X, Y, Z))


def arith_bases(n, a0=0x1234567, d=0xabcdef1):
    """n distinct bases with known discrete logs: P_i = [a0 + i*d] G.
    Returns (points_affine, dlogs).  O(n) group additions."""
    P = EC.jac_from_affine(EC.scalar_mul(EC.G1_GEN, a0))
    D = EC.jac_from_affine(EC.scalar_mul(EC.G1_GEN, d))
    jac = []
    for _ in range(n):
        jac.append(P)
        P = EC.jac_add(P, D)
    # batch-normalise
    zs = [p[2] for p in jac]
    zinv = F.batch_inverse(zs, F.Q_MOD)
    pts = []
    for (X, Y, Z), zi in zip(jac, zinv):
        zi2 = zi * zi % F.Q_MOD
        pts.append((X * zi2 % F.Q_MOD, Y * zi2 * zi % F.Q_MOD))
    dl = [(a0 + i * d) % F.R_MOD for i in range(n)]
    return pts, dl


def rand_fr(n, seed):
    rng = random.Random(seed)
    return [rng.randrange(F.R_MOD) for _ in range(n)]


def auto_window_bits(n):
    """The library's automatic window width for a base set of n points (capi.hip: auto_window_bits): lg(n) up to 2^18 points,
    lg(n) - 1 above, never 17 (it tiles 256 bits like 16 with twice the buckets), within [8, 20]."""
    lg = max(0, (int(n) - 1).bit_length())
    c = lg if lg <= 18 else lg - 1
    if c == 17:
        c = 16
    return min(20, max(8, c))


def hooks_env(env=None, **extra):
    """environment of a subprocess that loads libmarlin_hip[_bn254]_testhooks.so (the product's objects + the test hooks of
    include/marlin_hip_testhooks.h) instead of the product library, which exports no hook"""
    import os
    from marlin_amd import _lib
    e = dict(os.environ if env is None else env)
    e["MARLIN_AMD_LIB"] = _lib.HOOKS_LIB_PATH
    e.update(extra)
    return e
