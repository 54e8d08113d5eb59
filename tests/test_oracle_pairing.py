"""Pins for oracle/pairing.py (the BLS12-381 pairing behind the oracle's `PC::check_combinations`): the G2 generator is
on the twist and has order r; the map is non-degenerate, lands in the order-r subgroup of Fq12*, and is bilinear in
both arguments; a KZG10 opening checks through it (kzg10::check, [UPSTREAM-RECALLED])."""
import pytest
from oracle import fields as F

XI_A = 1 if F.CURVE == "bls12_381" else 9


def test_g2_generator_and_field_tower():
    from oracle import pairing as PR
    assert PR.g2_is_on_curve(PR.G2_GEN)
    assert PR.g2_mul(PR.G2_GEN, F.R_MOD - 1) == PR.g2_neg(PR.G2_GEN)
    a = PR.f12([3, 1, 4, 1, 5, 9, 2, 6, 5, 3, 5, 8])
    assert PR.f12_mul(a, PR.f12_inv(a)) == PR.F12_ONE
    u = PR.f12_from_fq2(0, 1)
    assert PR.f12_mul(u, u) == PR.f12([-1])                         # u^2 = -1
    w = PR.f12([0, 1])
    assert PR.f12_pow(w, 6) == PR.f12_from_fq2(XI_A, 1)             # w^6 = xi = XI_A + u


def test_pairing_is_bilinear_and_nondegenerate():
    from oracle import pairing as PR, curve as EC
    e = PR.pairing(EC.G1_GEN, PR.G2_GEN)
    assert e != PR.F12_ONE
    assert PR.f12_pow(e, F.R_MOD) == PR.F12_ONE
    a, b = 0x1234567, 0x9abcdef01
    assert PR.pairing(EC.scalar_mul(EC.G1_GEN, a), PR.g2_mul(PR.G2_GEN, b)) == PR.f12_pow(e, a * b % F.R_MOD)
    assert PR.pairing(None, PR.G2_GEN) == PR.F12_ONE


def test_kzg_opening_checks_through_the_pairing():
    """commit p(X), open at z: e(C - [v]G, H) == e(W, [tau]H - [z]H) with W = [q(tau)]G, q = (p - v) / (X - z)."""
    from oracle import pairing as PR, curve as EC
    from oracle.poly import poly_eval, divide_by_linear
    R = F.R_MOD
    tau, z = 0xdeadbeefcafe, 0x1337
    p = [5, 0, 7, 11, 13]
    v = poly_eval(p, z)
    q = divide_by_linear(p, z)
    C = EC.scalar_mul(EC.G1_GEN, poly_eval(p, tau))
    W = EC.scalar_mul(EC.G1_GEN, poly_eval(q, tau))
    h = PR.G2_GEN
    inner = PR.g2_add(PR.g2_mul(h, tau), PR.g2_neg(PR.g2_mul(h, z)))
    lhs = EC.add(C, EC.neg(EC.scalar_mul(EC.G1_GEN, v)))
    assert PR.pairing_product_is_one([(lhs, h), (EC.neg(W), inner)])
    assert not PR.pairing_product_is_one([(EC.add(lhs, EC.G1_GEN), h), (EC.neg(W), inner)])
