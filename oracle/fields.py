"""BLS12-381 field constants and Montgomery helpers (pure Python ints).

Oracle only (see oracle/__init__.py).  Restates the published parameters of
ark-bls12-381 0.3 (third-party dependency of the reference, Cargo.toml:37; not
vendored) -- SURVEY.md Appendix D re-derives every constant; tests/test_oracle_fields.py
re-checks them (primality, orders, Montgomery constants).
"""

import os

# Curve selection happens once, at import (ORACLE_CURVE=bls12_381 | bn254); the tests for the second curve
# (BASELINE.json configs[4]) run in a subprocess with ORACLE_CURVE=bn254.
CURVE = os.environ.get("ORACLE_CURVE", "bls12_381")

if CURVE == "bls12_381":
    # ---- scalar field Fr (ark_bls12_381::Fr; used at src/test.rs:120) -------------
    R_MOD = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
    FR_BITS = 255
    FR_TWO_ADICITY = 32
    FR_GENERATOR = 7
    FR_REPR_SHAVE_BITS = 1
    # ---- base field Fq -------------------------------------------------------------
    Q_MOD = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
    FQ_BITS = 381
    FQ_LIMBS64 = 6
    # ---- G1: y^2 = x^3 + 4 -----------------------------------------------------------
    G1_B = 4
    G1_GEN_X = 0x17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb
    G1_GEN_Y = 0x08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1
elif CURVE == "bn254":
    # BN254 (ark-bn254; not a dependency of the reference -- SURVEY.md Appendix E-2 -- public parameters,
    # SURVEY.md Appendix D): y^2 = x^3 + 3, generator (1, 2), Fr two-adicity 28, multiplicative generator 5.
    R_MOD = 21888242871839275222246405745257275088548364400416034343698204186575808495617
    FR_BITS = 254
    FR_TWO_ADICITY = 28
    FR_GENERATOR = 5
    FR_REPR_SHAVE_BITS = 2
    Q_MOD = 21888242871839275222246405745257275088696311157297823662689037894645226208583
    FQ_BITS = 254
    FQ_LIMBS64 = 4
    G1_B = 3
    G1_GEN_X = 1
    G1_GEN_Y = 2
else:
    raise ValueError("unknown ORACLE_CURVE %r" % CURVE)

FR_LIMBS64 = 4
FQ_BYTES = 8 * FQ_LIMBS64
FQ_RBITS = 64 * FQ_LIMBS64
# 2^s-th primitive root of unity = g^((r-1)/2^s)
FR_TWO_ADIC_ROOT = pow(FR_GENERATOR, (R_MOD - 1) >> FR_TWO_ADICITY, R_MOD)
FR_MONT_R = (1 << 256) % R_MOD
FR_MONT_R2 = (FR_MONT_R * FR_MONT_R) % R_MOD
FR_MONT_RINV = pow(FR_MONT_R, -1, R_MOD)
FR_INV64 = (-pow(R_MOD, -1, 1 << 64)) % (1 << 64)
FR_INV32 = (-pow(R_MOD, -1, 1 << 32)) % (1 << 32)
FQ_MONT_R = (1 << FQ_RBITS) % Q_MOD
FQ_MONT_R2 = (FQ_MONT_R * FQ_MONT_R) % Q_MOD
FQ_MONT_RINV = pow(FQ_MONT_R, -1, Q_MOD)
FQ_INV64 = (-pow(Q_MOD, -1, 1 << 64)) % (1 << 64)
FQ_INV32 = (-pow(Q_MOD, -1, 1 << 32)) % (1 << 32)


def fr_to_mont(x):
    return (x * FR_MONT_R) % R_MOD


def fr_from_mont(x):
    return (x * FR_MONT_RINV) % R_MOD


def fq_to_mont(x):
    return (x * FQ_MONT_R) % Q_MOD


def fq_from_mont(x):
    return (x * FQ_MONT_RINV) % Q_MOD


def to_limbs64(x, n):
    """little-endian u64 limbs (arkworks BigInteger in-memory order)."""
    return [(x >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(n)]


def from_limbs64(limbs):
    v = 0
    for i, l in enumerate(limbs):
        v |= int(l) << (64 * i)
    return v


def root_of_unity(log_n):
    """group_gen of the radix-2 domain of size 2^log_n [UPSTREAM-RECALLED B-1]:
    TWO_ADIC_ROOT_OF_UNITY squared (TWO_ADICITY - log_n) times."""
    assert 0 <= log_n <= FR_TWO_ADICITY
    return pow(FR_TWO_ADIC_ROOT, 1 << (FR_TWO_ADICITY - log_n), R_MOD)


def batch_inverse(vals, p=R_MOD):
    """Montgomery-trick batch inversion; zeros left untouched (ark_ff::batch_inversion,
    call sites src/ahp/prover.rs:663, src/ahp/mod.rs:314)."""
    prod = []
    acc = 1
    for v in vals:
        if v % p != 0:
            acc = acc * v % p
        prod.append(acc)
    inv = pow(acc, -1, p)
    out = list(vals)
    for i in range(len(vals) - 1, -1, -1):
        if vals[i] % p == 0:
            continue
        prev = prod[i - 1] if i > 0 else 1
        out[i] = inv * prev % p
        inv = inv * vals[i] % p
    return out
