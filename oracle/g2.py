"""G2 (the sextic twist E'(Fq2): y^2 = x^3 + b', Fq2 = Fq[u] / (u^2 + 1)) for the oracle: affine arithmetic on Python
ints, BLS12-381 or BN254 by ORACLE_CURVE (test infrastructure only).

Restates what ark-ec / ark-bls12-381 / ark-bn254 0.3 provide for `G2Affine` (third-party, absent from /root/reference;
public parameters: the IETF pairing-friendly-curves draft for BLS12-381, EIP-197 for BN254).  Pinned by
tests/test_oracle_pairing.py: generator on the twist, [r]H = O, b' (9 + u) = 3 on BN254.
"""
from .fields import Q_MOD as P, R_MOD as R, CURVE


def f2_add(a, b): return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)
def f2_sub(a, b): return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)
def f2_neg(a): return ((-a[0]) % P, (-a[1]) % P)
def f2_mul(a, b): return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)
def f2_scale(a, k): return (a[0] * k % P, a[1] * k % P)


def f2_inv(a):
    n = pow(a[0] * a[0] + a[1] * a[1], -1, P)
    return (a[0] * n % P, (-a[1]) * n % P)


F2_ZERO, F2_ONE = (0, 0), (1, 0)

if CURVE == "bls12_381":
    G2_B = (4, 4)                                   # 4 (1 + u)
    # the standard generator of G2 (zkcrypto / IETF pairing-friendly-curves draft)
    G2_GEN = (
        (0x024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8,
         0x13e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e),
        (0x0ce5d527727d6e118cc9cdc6da2e351aadfd9baa8cbdd3a76d429a695160d12c923ac9cc3baca289e193548608b82801,
         0x0606c4a02ea734cc32acd2b02bc28b99cb3e287e85a763af267492ab572e99ab3f370d275cec1da1aaa9075ff05f79be),
    )
else:
    G2_B = f2_scale(f2_inv((9, 1)), 3)              # 3 / (9 + u)  (D-type twist)
    # EIP-197's generator of G2
    G2_GEN = (
        (10857046999023057135944570762232829481370756359578518086990519993285655852781,
         11559732032986387107991004021392285783925812861821192530917403151452391805634),
        (8495653923123431417604973247489272438418190587263600148770280649306958101930,
         4082367875863433681332203403145435568316851327593401208105741076214120093531),
    )


def g2_is_on_curve(pt):
    if pt is None:
        return True
    x, y = pt
    return f2_mul(y, y) == f2_add(f2_mul(f2_mul(x, x), x), G2_B)


def g2_add(p1, p2):
    if p1 is None:
        return p2
    if p2 is None:
        return p1
    (x1, y1), (x2, y2) = p1, p2
    if x1 == x2:
        if y1 != y2 or y1 == F2_ZERO:
            return None
        m = f2_mul(f2_scale(f2_mul(x1, x1), 3), f2_inv(f2_scale(y1, 2)))
    else:
        m = f2_mul(f2_sub(y2, y1), f2_inv(f2_sub(x2, x1)))
    x3 = f2_sub(f2_sub(f2_mul(m, m), x1), x2)
    return (x3, f2_sub(f2_mul(m, f2_sub(x1, x3)), y1))


def g2_neg(p):
    return None if p is None else (p[0], f2_neg(p[1]))


def g2_mul(p, k):
    k %= R
    out = None
    while k:
        if k & 1:
            out = g2_add(out, p)
        p = g2_add(p, p)
        k >>= 1
    return out


def g2_msm_naive(points, scalars):
    acc = None
    for pt, s in zip(points, scalars):
        acc = g2_add(acc, g2_mul(pt, s))
    return acc
