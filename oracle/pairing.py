"""BLS12-381 / BN254 pairing for the oracle's verifier (test infrastructure; pure Python ints, slow and obviously structured).

`Marlin::verify` ends in `PC::check_combinations` (/root/reference src/lib.rs:413-423), which for MarlinKZG10 /
SonicKZG10 is the KZG10 pairing equation of ark-poly-commit 0.3 `kzg10::check` (third-party, absent here; restated
from its published algorithm [UPSTREAM-RECALLED]):

    e(C - [v]G - [random_v] gamma_G,  H)  ==  e(W,  beta_H - [z]H)

The pairing itself is ark-ec's `Bls12::pairing` (optimal ate: Miller loop over |x| = 0xd201000000010000 followed by the
final exponentiation (p^12 - 1) / r).  This file evaluates the same bilinear map the textbook way:

    Fq12 = Fq[w] / (w^12 - 2 w^6 + 2)            (u = w^6 - 1 satisfies u^2 = -1; xi = 1 + u = w^6)
    G2   = E'(Fq2): y^2 = x^3 + 4 xi  (M-type sextic twist), untwisted into E(Fq12) by (x, y) -> (x / w^2, y / w^3)
    e(P, Q) = f_{|x|, Q}(P) ^ ((p^12 - 1) / r)   with affine line functions over Fq12

Any non-degenerate bilinear map decides the KZG equation identically, so the sign convention of x (arkworks conjugates
because x < 0) is immaterial for a verifier; what is pinned here is bilinearity and non-degeneracy
(tests/test_oracle_pairing.py: e([a]P, [b]Q) = e(P, Q)^(ab), e(P, Q) != 1, e(P, Q)^r = 1).

With ORACLE_CURVE=bn254 (BASELINE configs[4]) the same code evaluates the ate pairing of BN254:

    Fq12 = Fq[w] / (w^12 - 18 w^6 + 82)          (u = w^6 - 9, xi = 9 + u = w^6)
    G2   = E'(Fq2): y^2 = x^3 + 3 / xi  (D-type twist), untwisted by (x, y) -> (x w^2, y w^3)
    e(P, Q) = f_{t-1, Q}(P) ^ ((p^12 - 1) / r),  t - 1 = 6 x^2, x = 4965661367192848881

(ark-ec's `Bn::pairing` runs the shorter optimal-ate loop 6x + 2 with two Frobenius line steps; both are non-degenerate
bilinear maps G1 x G2 -> mu_r, which is all the KZG check uses.)
"""
from .fields import Q_MOD as P, R_MOD as R, CURVE

if CURVE == "bls12_381":
    ATE_LOOP_COUNT = 0xd201000000010000
    XI_A = 1                                           # xi = XI_A + u
    TWIST_M = True
else:
    ATE_LOOP_COUNT = 6 * 4965661367192848881 ** 2      # t - 1
    XI_A = 9
    TWIST_M = False
# w^6 = xi = XI_A + u and u^2 = -1  =>  (w^6 - XI_A)^2 = -1  =>  w^12 = 2 XI_A w^6 - (XI_A^2 + 1)
_RED6, _RED0 = 2 * XI_A, XI_A * XI_A + 1


# ---- Fq12 as polynomials of degree < 12 over Fq ------------------------------------------------------------------
def f12(coeffs):
    c = [x % P for x in coeffs]
    return tuple(c + [0] * (12 - len(c)))


F12_ONE = f12([1])
F12_ZERO = f12([0])


def f12_add(a, b):
    return tuple((x + y) % P for x, y in zip(a, b))


def f12_sub(a, b):
    return tuple((x - y) % P for x, y in zip(a, b))


def f12_neg(a):
    return tuple((-x) % P for x in a)


def f12_scale(a, k):
    return tuple(x * k % P for x in a)


def f12_mul(a, b):
    t = [0] * 23
    for i, x in enumerate(a):
        if x:
            for j, y in enumerate(b):
                t[i + j] += x * y
    # reduce: w^12 = _RED6 w^6 - _RED0
    for k in range(22, 11, -1):
        v = t[k]
        if v:
            t[k - 6] += _RED6 * v
            t[k - 12] -= _RED0 * v
    return tuple(x % P for x in t[:12])


def f12_sqr(a):
    return f12_mul(a, a)


def _poly_deg(p):
    d = len(p) - 1
    while d and p[d] == 0:
        d -= 1
    return d


def _poly_rounded_div(a, b):
    dega, degb = _poly_deg(a), _poly_deg(b)
    temp = list(a)
    o = [0] * len(a)
    binv = pow(b[degb], -1, P)
    for i in range(dega - degb, -1, -1):
        q = temp[degb + i] * binv % P
        o[i] = (o[i] + q) % P
        for c in range(degb + 1):
            temp[c + i] = (temp[c + i] - q * b[c]) % P
    return o[:_poly_deg(o) + 1]


def f12_inv(a):
    """extended Euclid over Fq[w] against the modulus w^12 - _RED6 w^6 + _RED0"""
    lm, hm = [1] + [0] * 12, [0] * 13
    low, high = list(a) + [0], [_RED0, 0, 0, 0, 0, 0, (-_RED6) % P, 0, 0, 0, 0, 0, 1]
    while _poly_deg(low):
        r = _poly_rounded_div(high, low)
        r += [0] * (13 - len(r))
        nm, new = list(hm), list(high)
        for i in range(13):
            for j in range(13 - i):
                nm[i + j] = (nm[i + j] - lm[i] * r[j]) % P
                new[i + j] = (new[i + j] - low[i] * r[j]) % P
        lm, low, hm, high = nm, new, lm, low
    inv0 = pow(low[0], -1, P)
    return tuple(x * inv0 % P for x in lm[:12])


def f12_pow(a, e):
    out, base = F12_ONE, a
    while e:
        if e & 1:
            out = f12_mul(out, base)
        base = f12_sqr(base)
        e >>= 1
    return out


def f12_from_fq2(c0, c1):
    """a + b u with u = w^6 - XI_A"""
    return f12([(c0 - XI_A * c1) % P, 0, 0, 0, 0, 0, c1 % P])


# ---- Fq2 and G2 arithmetic on the twist: oracle/g2.py ---------------------------------------------------------------
from .g2 import (f2_add, f2_sub, f2_neg, f2_mul, f2_scale, f2_inv, F2_ZERO, F2_ONE, G2_B, G2_GEN,    # noqa: E402,F401
                 g2_is_on_curve, g2_add, g2_neg, g2_mul)


# ---- the pairing ----------------------------------------------------------------------------------------------------
_W = f12([0, 1])
_W2, _W3 = f12_mul(_W, _W), f12_mul(f12_mul(_W, _W), _W)
_W2_INV, _W3_INV = f12_inv(_W2), f12_inv(_W3)


def _untwist(q):
    (x0, x1), (y0, y1) = q
    if TWIST_M:
        return (f12_mul(f12_from_fq2(x0, x1), _W2_INV), f12_mul(f12_from_fq2(y0, y1), _W3_INV))
    return (f12_mul(f12_from_fq2(x0, x1), _W2), f12_mul(f12_from_fq2(y0, y1), _W3))


def _e12_double(p):
    x, y = p
    m = f12_mul(f12_scale(f12_sqr(x), 3), f12_inv(f12_scale(y, 2)))
    nx = f12_sub(f12_sqr(m), f12_scale(x, 2))
    return (nx, f12_sub(f12_mul(m, f12_sub(x, nx)), y)), m


def _e12_add(p, q):
    (x1, y1), (x2, y2) = p, q
    m = f12_mul(f12_sub(y2, y1), f12_inv(f12_sub(x2, x1)))
    nx = f12_sub(f12_sub(f12_sqr(m), x1), x2)
    return (nx, f12_sub(f12_mul(m, f12_sub(x1, nx)), y1)), m


def miller_loop(p_g1, q_g2):
    """f_{|x|, Q}(P) with P in E(Fq), Q in E'(Fq2); either argument None (identity) gives 1."""
    if p_g1 is None or q_g2 is None:
        return F12_ONE
    Q = _untwist(q_g2)
    px, py = f12([p_g1[0]]), f12([p_g1[1]])
    T = Q
    f = F12_ONE
    for i in range(ATE_LOOP_COUNT.bit_length() - 2, -1, -1):
        (nT, m) = _e12_double(T)
        line = f12_sub(f12_mul(m, f12_sub(px, T[0])), f12_sub(py, T[1]))       # tangent at T, evaluated at P
        f = f12_mul(f12_sqr(f), line)
        T = nT
        if (ATE_LOOP_COUNT >> i) & 1:
            (nT, m) = _e12_add(T, Q)
            line = f12_sub(f12_mul(m, f12_sub(px, T[0])), f12_sub(py, T[1]))   # chord through T and Q
            f = f12_mul(f, line)
            T = nT
    return f


FINAL_EXP = (P ** 12 - 1) // R


def final_exponentiation(f):
    return f12_pow(f, FINAL_EXP)


def pairing(p_g1, q_g2):
    return final_exponentiation(miller_loop(p_g1, q_g2))


def pairing_product_is_one(pairs):
    """prod_i e(P_i, Q_i) == 1  (one shared final exponentiation, like ark-ec's product_of_pairings)"""
    f = F12_ONE
    for p, q in pairs:
        f = f12_mul(f, miller_loop(p, q))
    return final_exponentiation(f) == F12_ONE
