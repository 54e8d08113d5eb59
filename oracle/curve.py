"""BLS12-381 G1 arithmetic, naive and windowed MSM (pure Python ints, canonical
(non-Montgomery) coordinates).  Oracle only (see oracle/__init__.py).

Restates short-Weierstrass group law of ark-ec 0.3 (transitive dependency of the
reference via ark-poly-commit, Cargo.toml:28; sources absent).  A point is
``None`` (identity) or an affine pair ``(x, y)`` of ints mod q.  MSM results are
unique group elements, so any correct algorithm pins any other.
"""
from .fields import Q_MOD as P, R_MOD, G1_GEN_X, G1_GEN_Y, G1_B

G1_GEN = (G1_GEN_X, G1_GEN_Y)


def is_on_curve(pt):
    if pt is None:
        return True
    x, y = pt
    return (y * y - x * x * x - G1_B) % P == 0


def neg(pt):
    if pt is None:
        return None
    return (pt[0], (-pt[1]) % P)


def add(a, b):
    if a is None:
        return b
    if b is None:
        return a
    x1, y1 = a
    x2, y2 = b
    if x1 == x2:
        if (y1 + y2) % P == 0:
            return None
        lam = (3 * x1 * x1) * pow(2 * y1, -1, P) % P
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, P) % P
    x3 = (lam * lam - x1 - x2) % P
    y3 = (lam * (x1 - x3) - y1) % P
    return (x3, y3)


# Jacobian (X, Y, Z), x = X/Z^2, y = Y/Z^3; Z = 0 is the identity -- the layout
# arkworks' GroupProjective uses and mh_msm returns.
def jac_from_affine(pt):
    if pt is None:
        return (1, 1, 0)
    return (pt[0], pt[1], 1)


def jac_to_affine(j):
    X, Y, Z = j
    if Z % P == 0:
        return None
    zi = pow(Z, -1, P)
    zi2 = zi * zi % P
    return (X * zi2 % P, Y * zi2 * zi % P)


def jac_double(j):
    X, Y, Z = j
    if Z % P == 0:
        return j
    A = X * X % P
    B = Y * Y % P
    C = B * B % P
    D = 2 * ((X + B) * (X + B) - A - C) % P
    E = 3 * A % P
    F = E * E % P
    X3 = (F - 2 * D) % P
    Y3 = (E * (D - X3) - 8 * C) % P
    Z3 = 2 * Y * Z % P
    return (X3, Y3, Z3)


def jac_add(a, b):
    X1, Y1, Z1 = a
    X2, Y2, Z2 = b
    if Z1 % P == 0:
        return b
    if Z2 % P == 0:
        return a
    Z1Z1 = Z1 * Z1 % P
    Z2Z2 = Z2 * Z2 % P
    U1 = X1 * Z2Z2 % P
    U2 = X2 * Z1Z1 % P
    S1 = Y1 * Z2 * Z2Z2 % P
    S2 = Y2 * Z1 * Z1Z1 % P
    if U1 == U2:
        if S1 == S2:
            return jac_double(a)
        return (1, 1, 0)
    H = (U2 - U1) % P
    Rr = (S2 - S1) % P
    HH = H * H % P
    HHH = H * HH % P
    V = U1 * HH % P
    X3 = (Rr * Rr - HHH - 2 * V) % P
    Y3 = (Rr * (V - X3) - S1 * HHH) % P
    Z3 = Z1 * Z2 * H % P
    return (X3, Y3, Z3)


def scalar_mul(pt, k):
    """double-and-add in Jacobian, returns affine."""
    k %= R_MOD
    acc = (1, 1, 0)
    base = jac_from_affine(pt)
    while k:
        if k & 1:
            acc = jac_add(acc, base)
        base = jac_double(base)
        k >>= 1
    return jac_to_affine(acc)


def msm_naive(bases, scalars):
    """sum_i scalars[i] * bases[i]; the definition VariableBaseMSM::multi_scalar_mul
    must equal (ark-ec; reached via PC::commit at src/lib.rs:172,193,213 and
    PC::open_combinations at src/lib.rs:292)."""
    acc = (1, 1, 0)
    for b, s in zip(bases, scalars):
        if s % R_MOD == 0 or b is None:
            continue
        acc = jac_add(acc, jac_from_affine(scalar_mul(b, s)))
    return jac_to_affine(acc)


def msm_pippenger(bases, scalars, c=None):
    """Bucket method with arkworks' window rule [UPSTREAM-RECALLED B-2]:
    c = 3 if n < 32 else ceil(log2 n)*69/100 + 2."""
    n = min(len(bases), len(scalars))
    if n == 0:
        return None
    if c is None:
        c = 3 if n < 32 else ((n - 1).bit_length() * 69 // 100) + 2
    nbits = 255
    total = (1, 1, 0)
    windows = list(range(0, nbits, c))
    for w in reversed(windows):
        for _ in range(c):
            total = jac_double(total)
        buckets = [(1, 1, 0)] * ((1 << c) - 1)
        for b, s in zip(bases[:n], scalars[:n]):
            d = ((s % R_MOD) >> w) & ((1 << c) - 1)
            if d and b is not None:
                buckets[d - 1] = jac_add(buckets[d - 1], jac_from_affine(b))
        running = (1, 1, 0)
        res = (1, 1, 0)
        for bk in reversed(buckets):
            running = jac_add(running, bk)
            res = jac_add(res, running)
        total = jac_add(total, res)
    return jac_to_affine(total)


def srs_powers(tau, n, base=G1_GEN):
    """[tau^i]base for i < n (KZG10::setup's powers_of_g [UPSTREAM-RECALLED B-3];
    known-tau so commitments are checkable in O(1): SURVEY.md §8c)."""
    out = []
    t = 1
    for _ in range(n):
        out.append(scalar_mul(base, t))
        t = t * tau % R_MOD
    return out
