"""Radix-2 evaluation domain + dense polynomial helpers over BLS12-381 Fr
(pure Python ints, canonical values).  Oracle only (see oracle/__init__.py).

Restates the behaviour of ark-poly 0.3 (`GeneralEvaluationDomain`,
`DensePolynomial`; third-party, Cargo.toml:26, not vendored) that the reference
calls at the sites listed in SURVEY.md Appendix A / B-1 / B-8.
"""
from .fields import R_MOD as R, root_of_unity, batch_inverse


def dft_naive(coeffs, log_n, inverse=False):
    """Definition: out[i] = sum_j coeffs[j] * w^(i*j) (natural order). O(n^2)."""
    n = 1 << log_n
    a = list(coeffs) + [0] * (n - len(coeffs))
    w = root_of_unity(log_n)
    if inverse:
        w = pow(w, -1, R)
    out = []
    for i in range(n):
        wi = pow(w, i, R)
        acc = 0
        x = 1
        for j in range(n):
            acc = (acc + a[j] * x) % R
            x = x * wi % R
        out.append(acc)
    if inverse:
        ninv = pow(n, -1, R)
        out = [v * ninv % R for v in out]
    return out


def ntt(vals, log_n, inverse=False):
    """Iterative radix-2 (bit-reverse + DIT), natural order in and out;
    inverse multiplies by n^-1 (Radix2EvaluationDomain::{fft,ifft}_in_place,
    [UPSTREAM-RECALLED B-1])."""
    n = 1 << log_n
    a = [v % R for v in vals] + [0] * (n - len(vals))
    assert len(a) == n
    for i in range(n):
        j = int(format(i, '0%db' % log_n)[::-1], 2) if log_n else 0
        if i < j:
            a[i], a[j] = a[j], a[i]
    w_n = root_of_unity(log_n)
    if inverse:
        w_n = pow(w_n, -1, R)
    m = 1
    while m < n:
        w_m = pow(w_n, n // (2 * m), R)
        for s in range(0, n, 2 * m):
            w = 1
            for k in range(m):
                t = a[s + k + m] * w % R
                u = a[s + k]
                a[s + k] = (u + t) % R
                a[s + k + m] = (u - t) % R
                w = w * w_m % R
        m *= 2
    if inverse:
        ninv = pow(n, -1, R)
        a = [v * ninv % R for v in a]
    return a


class Domain:
    """GeneralEvaluationDomain::new(k) -> radix-2 domain of size next_pow2(k)."""

    def __init__(self, min_size):
        size = 1
        log = 0
        while size < min_size:
            size *= 2
            log += 1
        self.size = size
        self.log_size = log
        self.group_gen = root_of_unity(log)
        self.group_gen_inv = pow(self.group_gen, -1, R)
        self.size_inv = pow(size, -1, R)

    def elements(self):
        out = []
        x = 1
        for _ in range(self.size):
            out.append(x)
            x = x * self.group_gen % R
        return out

    def element(self, i):
        return pow(self.group_gen, i, R)

    def fft(self, coeffs):
        assert len(coeffs) <= self.size
        return ntt(coeffs, self.log_size)

    def ifft(self, evals):
        assert len(evals) <= self.size
        return ntt(evals, self.log_size, inverse=True)

    def coset_fft(self, coeffs):
        """Radix2EvaluationDomain::coset_fft [ark-poly 0.3, UPSTREAM-RECALLED]: distribute_powers(coeffs, g) then fft --
        the evaluations of the polynomial on the coset g H, g = F::multiplicative_generator()."""
        from .fields import FR_GENERATOR as g
        assert len(coeffs) <= self.size
        return ntt([c * pow(g, i, R) % R for i, c in enumerate(coeffs)], self.log_size)

    def coset_ifft(self, evals):
        """Radix2EvaluationDomain::coset_ifft: ifft then distribute_powers(coeffs, g^-1)."""
        from .fields import FR_GENERATOR as g
        ginv = pow(g, -1, R)
        return [c * pow(ginv, i, R) % R for i, c in enumerate(self.ifft(evals))]

    def evaluate_vanishing_polynomial(self, tau):
        return (pow(tau, self.size, R) - 1) % R

    def reindex_by_subdomain(self, other, index):
        # [UPSTREAM-RECALLED B-1]; used at src/ahp/prover.rs:422,
        # src/ahp/constraint_systems.rs:180
        period = self.size // other.size
        if index < other.size:
            return index * period
        i = index - other.size
        x = period - 1
        return i + (i // x) + 1

    def evaluate_all_lagrange_coefficients(self, tau):
        """L_i(tau) for all i (used at src/ahp/mod.rs:154-159)."""
        z = self.evaluate_vanishing_polynomial(tau)
        els = self.elements()
        if z == 0:
            return [1 if e == tau % R else 0 for e in els]
        # L_i(tau) = z * w^i / (n * (tau - w^i))
        dens = batch_inverse([(self.size * (tau - e)) % R for e in els])
        return [z * e % R * d % R for e, d in zip(els, dens)]

    # --- UnnormalizedBivariateLagrangePoly (src/ahp/mod.rs:301-328) ---------------
    def eval_unnormalized_bivariate_lagrange_poly(self, x, y):
        if x % R != y % R:
            return ((self.evaluate_vanishing_polynomial(x) - self.evaluate_vanishing_polynomial(y))
                    * pow(x - y, -1, R)) % R
        return self.size * pow(x, self.size - 1, R) % R

    def batch_eval_unnormalized_bivariate_lagrange_poly_with_diff_inputs(self, x):
        vanish_x = self.evaluate_vanishing_polynomial(x)
        inv = batch_inverse([(x - y) % R for y in self.elements()])
        return [v * vanish_x % R for v in inv]

    def batch_eval_unnormalized_bivariate_lagrange_poly_with_same_inputs(self):
        elems = [e * self.size % R for e in self.elements()]
        return [elems[0]] + elems[1:][::-1]


# ---- DensePolynomial helpers (coefficient lists, low -> high) -------------------
def trim(c):
    """from_coefficients_vec strips trailing zeros [UPSTREAM-RECALLED B-8]."""
    c = list(c)
    while c and c[-1] % R == 0:
        c.pop()
    return c


def degree(c):
    c = trim(c)
    return len(c) - 1 if c else 0


def poly_add(a, b):
    n = max(len(a), len(b))
    return trim([((a[i] if i < len(a) else 0) + (b[i] if i < len(b) else 0)) % R for i in range(n)])


def poly_sub(a, b):
    n = max(len(a), len(b))
    return trim([((a[i] if i < len(a) else 0) - (b[i] if i < len(b) else 0)) % R for i in range(n)])


def poly_scale(a, s):
    return trim([v * s % R for v in a])


def poly_mul(a, b):
    """`&a * &b` via FFT on the domain of size np2(len_a + len_b - 1) [B-8]."""
    a = trim(a)
    b = trim(b)
    if not a or not b:
        return []
    d = Domain(len(a) + len(b) - 1)
    ea = d.fft(a)
    eb = d.fft(b)
    return trim(d.ifft([x * y % R for x, y in zip(ea, eb)]))


def poly_eval(c, x):
    acc = 0
    for v in reversed(c):
        acc = (acc * x + v) % R
    return acc


def mul_by_vanishing_poly(c, domain_size):
    """p(X) * (X^n - 1) [B-8] (src/ahp/prover.rs:512)."""
    out = [0] * domain_size + list(c)
    for i, v in enumerate(c):
        out[i] = (out[i] - v) % R
    return trim(out)


def divide_by_vanishing_poly(c, domain_size):
    """(q, r) with p = q*(X^n - 1) + r [B-8] (src/ahp/prover.rs:353,550,686)."""
    c = trim(c)
    n = domain_size
    if len(c) < n:
        return [], c
    # q_i = sum_{j>=1} p_{i + j n}: a suffix sum along every residue class mod n (q_i = p_{i+n} + q_{i+n}); r_i = p_i + q_i
    q = [0] * (len(c) - n)
    for i in range(len(q) - 1, -1, -1):
        q[i] = (c[i + n] + (q[i + n] if i + n < len(q) else 0)) % R
    r = [(c[i] + (q[i] if i < len(q) else 0)) % R for i in range(n)]
    return trim(q), trim(r)


def divide_by_linear(c, z):
    """(p(X) - p(z)) / (X - z): KZG10::open's witness polynomial [B-4]."""
    c = trim(c)
    if len(c) <= 1:
        return []
    q = [0] * (len(c) - 1)
    acc = 0
    for i in range(len(c) - 1, 0, -1):
        acc = (acc * z + c[i]) % R
        q[i - 1] = acc
    return trim(q)
