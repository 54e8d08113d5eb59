"""Restatement of the reference's AHP for R1CS (oracle only; canonical Python ints).

Follows /root/reference:
  circuits          benches/bench.rs:45-66 (DummyCircuit), src/test.rs:16-50 (Circuit)
  padding / square  src/ahp/constraint_systems.rs:41-81
  indexer           src/ahp/indexer.rs:83-102,151-234; constraint_systems.rs:125-262
  prover rounds     src/ahp/prover.rs:211-306 (init), 309-409, 411-428, 443-570, 588-706
  verifier msgs     src/ahp/verifier.rs:44-188
  linear combos     src/ahp/mod.rs:110-221
Constraint synthesis itself (ark-relations, LC outlining) is out of scope:
circuits are given directly as matrices + assignments, in the column order
`ConstraintSystem::to_matrices` produces (instance variables first, then witnesses).
"""
from .fields import R_MOD as R, batch_inverse
from .poly import (Domain, trim, poly_add, poly_sub, poly_mul, poly_eval, poly_scale, mul_by_vanishing_poly,
                   divide_by_vanishing_poly, degree)
from .fs import fr_rand


# ----------------------------------------------------------------------------------
# circuits -> (instance assignment incl. One, witness assignment, A, B, C rows)
# ----------------------------------------------------------------------------------
class R1CS:
    def __init__(self, instance, witness, a, b, c):
        self.instance = list(instance)     # formatted: [1, public inputs...]
        self.witness = list(witness)
        self.a, self.b, self.c = a, b, c   # rows of [(coeff, column)]

    @property
    def num_constraints(self):
        return len(self.a)


def dummy_circuit(a, b, num_variables, num_constraints):
    """benches/bench.rs:45-66: witnesses a, b, (num_variables-3) copies of a; input c = a*b;
    num_constraints-1 rows a*b=c, one empty row."""
    inst = [1, a * b % R]
    wit = [a % R, b % R] + [a % R] * (num_variables - 3)
    ni = 2
    col_a, col_b, col_c = ni + 0, ni + 1, 1
    A = [[(1, col_a)] for _ in range(num_constraints - 1)] + [[]]
    B = [[(1, col_b)] for _ in range(num_constraints - 1)] + [[]]
    C = [[(1, col_c)] for _ in range(num_constraints - 1)] + [[]]
    return R1CS(inst, wit, A, B, C)


def test_circuit(a, b, num_constraints, num_variables):
    """src/test.rs:16-50: inputs c = ab, d = ab^2; num_constraints-1 rows a*b=c and one row c*b=d."""
    c = a * b % R
    d = c * b % R
    inst = [1, c, d]
    wit = [a % R, b % R] + [a % R] * (num_variables - 3)
    return ("unpadded", inst, wit, num_constraints)


def finalize_test_circuit(spec):
    _, inst, wit, num_constraints = spec
    # column indices depend on the padded number of instance variables (3 -> 4)
    ni = 1
    while ni < len(inst):
        ni *= 2
    col_a, col_b, col_c, col_d = ni + 0, ni + 1, 1, 2
    A = [[(1, col_a)] for _ in range(num_constraints - 1)] + [[(1, col_c)]]
    B = [[(1, col_b)] for _ in range(num_constraints - 1)] + [[(1, col_b)]]
    C = [[(1, col_c)] for _ in range(num_constraints - 1)] + [[(1, col_d)]]
    return R1CS(inst, wit, A, B, C)


def pad_and_square(cs):
    """pad_input_for_indexer_and_prover (constraint_systems.rs:45-58) + make_matrices_square
    (60-81).  Column indices in cs.a/b/c must already assume the padded instance count."""
    ni = 1
    while ni < len(cs.instance):
        ni *= 2
    inst = cs.instance + [0] * (ni - len(cs.instance))
    wit = list(cs.witness)
    a, b, c = [list(r) for r in cs.a], [list(r) for r in cs.b], [list(r) for r in cs.c]
    num_vars = len(inst) + len(wit)
    nc = len(a)
    if num_vars > nc:
        for _ in range(num_vars - nc):
            a.append([]); b.append([]); c.append([])
    else:
        wit += [1] * (nc - num_vars)
    out = R1CS(inst, wit, a, b, c)
    assert len(out.instance) + len(out.witness) == out.num_constraints
    return out


# ----------------------------------------------------------------------------------
# indexer
# ----------------------------------------------------------------------------------
class Index:
    pass


def max_degree(num_constraints, num_variables, num_non_zero):
    """src/ahp/mod.rs:71-93."""
    h = Domain(max(num_variables, num_constraints)).size
    k = Domain(num_non_zero).size
    return max(2 * h + 1 - 2, 3 * h + 2 - 3, h, h, k - 1)


def index(cs):
    """AHPForR1CS::index (indexer.rs:151-234) on an already padded/squared system."""
    idx = Index()
    a, b, c = cs.a, cs.b, cs.c
    joint = [sorted(set([j for _, j in ra] + [j for _, j in rb] + [j for _, j in rc])) for ra, rb, rc in zip(a, b, c)]
    nnz = sum(len(r) for r in joint)
    idx.num_variables = len(cs.instance) + len(cs.witness)
    idx.num_constraints = cs.num_constraints
    idx.num_non_zero = nnz
    idx.num_instance_variables = len(cs.instance)
    idx.a, idx.b, idx.c = a, b, c
    dh, dk, dx = Domain(idx.num_constraints), Domain(nnz), Domain(idx.num_instance_variables)
    idx.domain_h, idx.domain_k, idx.domain_x = dh, dk, dx
    # arithmetize_matrix (constraint_systems.rs:125-262)
    elems = dh.elements()
    eq_vals = dict(zip(elems, dh.batch_eval_unnormalized_bivariate_lagrange_poly_with_same_inputs()))
    am = [dict(((r, j), f) for r, row in enumerate(m) for f, j in row) for m in (a, b, c)]
    row_vec, col_vec, va, vb, vc, inv = [], [], [], [], [], []
    for r, row in enumerate(joint):
        for i in row:
            row_val = elems[r]
            col_val = elems[dh.reindex_by_subdomain(dx, i)]
            row_vec.append(col_val)      # transpose
            col_vec.append(row_val)
            va.append(am[0].get((r, i), 0)); vb.append(am[1].get((r, i), 0)); vc.append(am[2].get((r, i), 0))
            inv.append(eq_vals[col_val])
    inv = batch_inverse(inv)
    va = [x * y % R for x, y in zip(va, inv)]
    vb = [x * y % R for x, y in zip(vb, inv)]
    vc = [x * y % R for x, y in zip(vc, inv)]
    pad = dk.size - len(row_vec)
    row_vec += [elems[0]] * pad; col_vec += [elems[0]] * pad
    va += [0] * pad; vb += [0] * pad; vc += [0] * pad
    rc_vec = [x * y % R for x, y in zip(row_vec, col_vec)]
    idx.evals_on_K = {"row": row_vec, "col": col_vec, "row_col": rc_vec, "val_a": va, "val_b": vb, "val_c": vc}
    idx.polys = {  # labels of INDEXER_POLYNOMIALS (mod.rs:33-36), coefficient form (trailing zeros stripped)
        "row": trim(dk.ifft(row_vec)), "col": trim(dk.ifft(col_vec)), "a_val": trim(dk.ifft(va)),
        "b_val": trim(dk.ifft(vb)), "c_val": trim(dk.ifft(vc)), "row_col": trim(dk.ifft(rc_vec))}
    idx.max_degree = max_degree(idx.num_constraints, idx.num_variables, nnz)
    return idx


INDEXER_POLYNOMIALS = ["row", "col", "a_val", "b_val", "c_val", "row_col"]


# ----------------------------------------------------------------------------------
# prover
# ----------------------------------------------------------------------------------
class ProverState:
    pass


def prover_init(idx, cs):
    """prover.rs:211-306 (synthesis replaced by the given assignment)."""
    st = ProverState()
    st.index = idx
    st.x = list(cs.instance)
    st.w = list(cs.witness)
    assert idx.num_constraints == cs.num_constraints and len(st.x) + len(st.w) == idx.num_variables
    z = st.x + st.w

    def mv(m):
        return [sum(f * z[j] for f, j in row) % R for row in m]
    st.z_a, st.z_b = mv(idx.a), mv(idx.b)
    st.domain_h, st.domain_k, st.domain_x = idx.domain_h, idx.domain_k, idx.domain_x
    st.zk_bound = 1
    return st


def public_input(st):
    return st.x[1:]          # unformat_public_input (constraint_systems.rs:278-280)


def prover_first_round(st, rng):
    """prover.rs:309-409.  rng draws: w blind, z_a blind, z_b blind, then 3H mask coefficients."""
    dh, dx = st.domain_h, st.domain_x
    H, X = dh.size, dx.size
    x_poly = trim(dx.ifft(st.x))
    x_evals = dh.fft(x_poly)
    ratio = H // X
    w_ext = st.w + [0] * (H - X - len(st.w))
    w_evals = [0 if k % ratio == 0 else (w_ext[k - (k // ratio) - 1] - x_evals[k]) % R for k in range(H)]
    v_H = [R - 1] + [0] * (H - 1) + [1]

    def blind(evals):
        r = fr_rand(rng)
        return poly_add(trim(dh.ifft(evals)), poly_mul([r], v_H))
    w_poly = blind(w_evals)
    w_poly, rem = divide_by_vanishing_poly(w_poly, X)
    assert not rem
    z_a_poly = blind(st.z_a)
    z_b_poly = blind(st.z_b)
    mask_deg = 3 * H + 2 * st.zk_bound - 3
    mask = [fr_rand(rng) for _ in range(mask_deg + 1)]
    r0 = sum(mask[H * i] for i in range(mask_deg // H + 1)) % R
    mask[0] = (mask[0] - r0) % R
    mask = trim(mask)
    assert degree(w_poly) < H - X + st.zk_bound and degree(z_a_poly) < H + st.zk_bound
    st.w_poly, st.z_a_poly, st.z_b_poly, st.mask_poly = w_poly, z_a_poly, z_b_poly, mask
    # (label, poly, degree_bound, hiding_bound)  prover.rs:390-394
    return [("w", w_poly, None, 1), ("z_a", z_a_poly, None, 1), ("z_b", z_b_poly, None, 1), ("mask_poly", mask, None, None)]


def calculate_t(idx, etas, r_alpha_x_on_h):
    """prover.rs:411-428."""
    dh, dx = idx.domain_h, idx.domain_x
    t = [0] * dh.size
    for m, eta in zip((idx.a, idx.b, idx.c), etas):
        for r, row in enumerate(m):
            for coeff, c in row:
                k = dh.reindex_by_subdomain(dx, c)
                t[k] = (t[k] + eta * coeff % R * r_alpha_x_on_h[r]) % R
    return trim(dh.ifft(t))


def prover_second_round(st, alpha, eta_a, eta_b, eta_c):
    """prover.rs:443-570."""
    dh, dx = st.domain_h, st.domain_x
    H = dh.size
    z_c = poly_mul(st.z_a_poly, st.z_b_poly)
    summed = [c * eta_c % R for c in z_c]
    for i, (a, b) in enumerate(zip(st.z_a_poly, st.z_b_poly)):
        if i < len(summed):
            summed[i] = (summed[i] + eta_a * a + eta_b * b) % R
    summed = trim(summed)
    r_alpha_evals = dh.batch_eval_unnormalized_bivariate_lagrange_poly_with_diff_inputs(alpha)
    r_alpha_poly = trim(dh.ifft(r_alpha_evals))
    t_poly = calculate_t(st.index, (eta_a, eta_b, eta_c), r_alpha_evals)
    x_poly = trim(dx.ifft(st.x))
    z_poly = mul_by_vanishing_poly(st.w_poly, dx.size)
    z_poly = z_poly + [0] * max(0, len(x_poly) - len(z_poly))
    for i, xv in enumerate(x_poly):
        z_poly[i] = (z_poly[i] + xv) % R
    assert degree(z_poly) < H + st.zk_bound
    mul_size = max(len(st.mask_poly), len(r_alpha_poly) + len(summed), len(t_poly) + len(z_poly))
    dm = Domain(mul_size)
    ra, sz, ze, te = dm.fft(r_alpha_poly), dm.fft(summed), dm.fft(z_poly), dm.fft(t_poly)
    rhs = trim(dm.ifft([(a * b - c * d) % R for a, b, c, d in zip(ra, sz, ze, te)]))
    q_1 = poly_add(st.mask_poly, rhs)
    h_1, x_g_1 = divide_by_vanishing_poly(q_1, H)
    g_1 = trim(x_g_1[1:])
    assert degree(g_1) <= H - 2 and degree(h_1) <= 2 * H + 2 * st.zk_bound - 2
    st.t_poly, st.g_1, st.h_1 = t_poly, g_1, h_1
    st.first_msg = (alpha, eta_a, eta_b, eta_c)
    return [("t", t_poly, None, None), ("g_1", g_1, H - 2, 1), ("h_1", h_1, None, None)]


def prover_third_round(st, beta):
    """prover.rs:588-706."""
    idx = st.index
    dh, dk = st.domain_h, st.domain_k
    alpha, eta_a, eta_b, eta_c = st.first_msg
    v = dh.evaluate_vanishing_polynomial(alpha) * dh.evaluate_vanishing_polynomial(beta) % R
    ea, eb, ec = eta_a * v % R, eta_b * v % R, eta_c * v % R
    pa, pb, pc = idx.polys["a_val"], idx.polys["b_val"], idx.polys["c_val"]
    n = min(len(pa), len(pb), len(pc))      # the reference zips the three coefficient vectors
    a_poly = trim([(ea * pa[i] + eb * pb[i] + ec * pc[i]) % R for i in range(n)])
    ev = idx.evals_on_K
    ab = alpha * beta % R
    b_evals = [(ab - alpha * r - beta * c + rc) % R for r, c, rc in zip(ev["row"], ev["col"], ev["row_col"])]
    b_poly = trim(dk.ifft(b_evals))
    inv = batch_inverse([(beta - r) * (alpha - c) % R for r, c in zip(ev["row"], ev["col"])])
    f_evals = [i * ((ea * x + eb * y + ec * z) % R) % R for i, x, y, z in zip(inv, ev["val_a"], ev["val_b"], ev["val_c"])]
    f = trim(dk.ifft(f_evals))
    h_2, _ = divide_by_vanishing_poly(poly_sub(a_poly, poly_mul(b_poly, f)), dk.size)
    g_2 = trim(f[1:])
    assert degree(h_2) <= dk.size - 2 and degree(g_2) <= dk.size - 2
    st.g_2, st.h_2 = g_2, h_2
    return [("g_2", g_2, dk.size - 2, None), ("h_2", h_2, None, None)]


# ----------------------------------------------------------------------------------
# verifier messages (drawn from the Fiat-Shamir rng inside prove) and linear combinations
# ----------------------------------------------------------------------------------
def sample_outside_domain(domain, fs):
    t = fs.rand_fr()
    while domain.evaluate_vanishing_polynomial(t) == 0:
        t = fs.rand_fr()
    return t


def verifier_first_round(dh, fs):           # verifier.rs:44-79
    alpha = sample_outside_domain(dh, fs)
    return alpha, fs.rand_fr(), fs.rand_fr(), fs.rand_fr()


def verifier_second_round(dh, fs):          # verifier.rs:82-91
    return sample_outside_domain(dh, fs)


def verifier_third_round(fs):               # verifier.rs:94-100
    return fs.rand_fr()


def query_set(beta, gamma):
    """verifier.rs:103-188; BTreeSet order of (label, (point_label, point))."""
    return sorted([("g_1", "beta", beta), ("z_b", "beta", beta), ("t", "beta", beta), ("outer_sumcheck", "beta", beta),
                   ("g_2", "gamma", gamma), ("inner_sumcheck", "gamma", gamma)])


def construct_linear_combinations(pub_in, evals, dh, dk, msgs):
    """src/ahp/mod.rs:110-221.  `evals(label, point)` returns the evaluation of a prover/indexer
    polynomial.  Returns {lc_label: [(coeff, poly_label or None for LCTerm::One)]}, sorted by label."""
    alpha, eta_a, eta_b, eta_c, beta, gamma = msgs
    x = [1] + list(pub_in)
    dx = Domain(len(x))
    r_ab = dh.eval_unnormalized_bivariate_lagrange_poly(alpha, beta)
    vHa = dh.evaluate_vanishing_polynomial(alpha)
    vHb = dh.evaluate_vanishing_polynomial(beta)
    vXb = dx.evaluate_vanishing_polynomial(beta)
    z_b_b, t_b, g_1_b = evals("z_b", beta), evals("t", beta), evals("g_1", beta)
    x_b = sum(l * xi for l, xi in zip(dx.evaluate_all_lagrange_coefficients(beta), x)) % R
    outer = [(1, "mask_poly"), (r_ab * (eta_a + eta_c * z_b_b) % R, "z_a"), (r_ab * eta_b % R * z_b_b % R, None),
             ((-t_b * vXb) % R, "w"), ((-t_b * x_b) % R, None), ((-vHb) % R, "h_1"), ((-beta * g_1_b) % R, None)]
    g_2_g = evals("g_2", gamma)
    vKg = dk.evaluate_vanishing_polynomial(gamma)
    va = vHa * vHb % R
    a_lc = [(eta_a * va % R, "a_val"), (eta_b * va % R, "b_val"), (eta_c * va % R, "c_val")]
    mult = (gamma * g_2_g + t_b * pow(dk.size, -1, R)) % R
    b_lc = [(beta * alpha % R * mult % R, None), ((-alpha) * mult % R, "row"), ((-beta) * mult % R, "col"), (mult, "row_col")]
    inner = list(a_lc) + [((-c) % R, l) for c, l in b_lc] + [((-vKg) % R, "h_2")]
    lcs = {"z_b": [(1, "z_b")], "g_1": [(1, "g_1")], "t": [(1, "t")], "outer_sumcheck": outer,
           "g_2": [(1, "g_2")], "inner_sumcheck": inner}
    return dict(sorted(lcs.items()))
