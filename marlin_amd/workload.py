"""Shapes of the Marlin::prove hot path for the reference's benchmark circuit.

`DummyCircuit` (/root/reference benches/bench.rs:26-66) with `num_variables = 10`
and N constraints gives |H| = N, |K| = 4N, |X| = 2.  One
`Marlin::prove` (src/lib.rs:151-311) with MarlinKZG10 then performs exactly the
transforms and multi-scalar multiplications listed in SURVEY.md Appendix A; this
module restates that inventory as data so bench.py, the tests and the CPU
baseline all run the same list.
"""


def ntt_inventory(H, K=None, X=2):
    """[(log2 size, inverse?, reference site)] for one prove."""
    K = 4 * H if K is None else K

    def lg(n):
        l = n.bit_length() - 1
        assert 1 << l == n
        return l
    inv = []
    inv += [(lg(X), True, "prover.rs:321-325 x_poly"), (lg(H), False, "prover.rs:326 x_evals")]
    for name, site in (("w", "350-352"), ("z_a", "359-360"), ("z_b", "365-366")):
        inv += [(lg(H), True, "prover.rs:%s %s" % (site, name))]
        inv += [(lg(2 * H), False, "prover.rs:%s r*v_H" % site), (lg(2 * H), False, "prover.rs:%s r*v_H" % site),
                (lg(2 * H), True, "prover.rs:%s r*v_H" % site)]
    inv += [(lg(4 * H), False, "prover.rs:467 z_a*z_b"), (lg(4 * H), False, "prover.rs:467 z_a*z_b"),
            (lg(4 * H), True, "prover.rs:467 z_a*z_b")]
    inv += [(lg(H), True, "prover.rs:488 r_alpha"), (lg(H), True, "prover.rs:427 t"), (lg(X), True, "prover.rs:506-510 x_poly")]
    inv += [(lg(4 * H), False, "prover.rs:532-535 q_1 evals")] * 4 + [(lg(4 * H), True, "prover.rs:545 rhs")]
    inv += [(lg(K), True, "prover.rs:655 b"), (lg(K), True, "prover.rs:681 f")]
    inv += [(lg(2 * K), False, "prover.rs:685 b*f"), (lg(2 * K), False, "prover.rs:685 b*f"), (lg(2 * K), True, "prover.rs:685 b*f")]
    assert len(inv) == 30
    return inv


def ntt_input_lengths(H, K=None, X=2):
    """For each transform of ntt_inventory, how many elements the caller's Vec holds when it reaches ark-poly's `fft_in_place` /
    `ifft_in_place` -- BEFORE their `resize(self.size(), zero)` -- i.e. what the seam has to upload (mh_ntt_len):
    `DensePolynomial * DensePolynomial` evaluates both factors on np2(len_a + len_b - 1) points (prover.rs:352,360,366: a constant
    times v_H of H + 1 coefficients; :467 z_a z_b, H + 1 each; :685 b f, K each), `evaluate_over_domain_by_ref` the q_1 factors
    on 4H points (:532-535: r_alpha H, summed_z_m 2H + 1, z H + 1, t H); inverse transforms arrive full."""
    K = 4 * H if K is None else K
    L = [X, X]
    for _ in range(3):
        L += [H, 1, H + 1, 2 * H]
    L += [H + 1, H + 1, 4 * H]
    L += [H, H, X]
    L += [H, 2 * H + 1, H + 1, H, 4 * H]
    L += [K, K]
    L += [K, K, 2 * K]
    inv = ntt_inventory(H, K, X)
    assert len(L) == len(inv) and all(l <= (1 << lg) for l, (lg, _, _) in zip(L, inv))
    return L


def ntt_executed(H, K=None, X=2):
    """[(log2 size, inverse?, what)]: the transforms mh_marlin_prove actually runs for the same proof (16 of the 30).
    Not executed, with identical outputs: the nine 2H transforms behind `const * v_H` (the product is known in closed
    form); the inverse of z_a*z_b and the forward transform of summed_z_m on the same 4H domain (they cancel: the
    evaluations are eta_c z_a z_b + eta_a z_a + eta_b z_b pointwise); the interpolation of b and the three 2K transforms
    of b*f -- h_2 is evaluated on the coset g K instead (one forward and one inverse transform of size K); the H-point
    evaluation of x_poly when the public input has at most 16 elements (Horner per point)."""
    K = 4 * H if K is None else K
    lg = lambda n: n.bit_length() - 1
    ex = [(lg(X), True, "x_poly")] + ([(lg(H), False, "x_evals")] if X > 16 else [])
    ex += [(lg(H), True, "w"), (lg(H), True, "z_a"), (lg(H), True, "z_b")]
    ex += [(lg(4 * H), False, "z_a on 4H"), (lg(4 * H), False, "z_b on 4H"), (lg(H), True, "r_alpha"), (lg(H), True, "t"),
           (lg(X), True, "x_poly")]
    ex += [(lg(4 * H), False, "r_alpha on 4H"), (lg(4 * H), False, "z on 4H"), (lg(4 * H), False, "t on 4H"), (lg(4 * H), True, "rhs")]
    ex += [(lg(K), True, "f"), (lg(K), False, "f on the coset g K"), (lg(K), True, "h_2 from the coset g K")]
    assert len(ex) == (17 if X > 16 else 16)
    return ex


def msm_inventory(H, K=None, X=2):
    """[(n_pairs, reference site)] of the large MSMs of one prove (MarlinKZG10); the
    3-coefficient hiding MSMs on powers_of_gamma_g are listed separately."""
    K = 4 * H if K is None else K
    big = [
        (H - X + 1, "lib.rs:172 w"), (H + 1, "lib.rs:172 z_a"), (H + 1, "lib.rs:172 z_b"), (3 * H, "lib.rs:172 mask_poly"),
        (H, "lib.rs:193 t"), (H - 1, "lib.rs:193 g_1"), (H - 1, "lib.rs:193 g_1 shifted"), (2 * H, "lib.rs:193 h_1"),
        (K - 1, "lib.rs:213 g_2"), (K - 1, "lib.rs:213 g_2 shifted"), (K - 1, "lib.rs:213 h_2"),
        (3 * H - 1, "lib.rs:292 open@beta witness"), (H - 2, "lib.rs:292 open@beta shifted witness (g_1)"),
        (K - 1, "lib.rs:292 open@gamma witness"), (K - 2, "lib.rs:292 open@gamma shifted witness (g_2)"),
    ]
    small = [(3, "lib.rs:172 w hiding"), (3, "lib.rs:172 z_a hiding"), (3, "lib.rs:172 z_b hiding"),
             (3, "lib.rs:193 g_1 hiding"), (3, "lib.rs:193 g_1 shifted hiding"), (2, "lib.rs:292 open@beta hiding")]
    return big, small


def msm_executed(H, K=None, X=2, pc="marlin"):
    """[(n_pairs, what)]: the large MSMs mh_marlin_prove runs (MarlinKZG10; pc="sonic": SonicKZG10, one commitment per
    polynomial -- against the shifted powers when degree-bounded -- and one witness MSM per opening point).  The 11 commitments are those of
    msm_inventory; each opening proof is ONE MSM, because the reference's `w + shifted_w` is the multi-scalar product
    of the witness coefficients plus the shifted witness's coefficients at offset max_degree - bound on the same SRS
    (at gamma the offset is 1 for this circuit, so 2 x 4H pairs become 4H)."""
    K = 4 * H if K is None else K
    big, _ = msm_inventory(H, K, X)
    if pc == "sonic":
        ex = [b for b in big[:11] if "shifted" not in b[1]]
        return ex + [(3 * H - 1, "open@beta witness"), (K - 1, "open@gamma witness")]
    D = max(3 * H - 1, K - 1)                     # AHPForR1CS::max_degree for this shape (mod.rs:71-93, zk_bound 1)
    ex = big[:11]
    ex.append((max(3 * H - 1, D - (H - 2) + H - 2), "lib.rs:292 open@beta witness + shifted witness (g_1)"))
    ex.append((max(K - 1, D - (K - 2) + K - 2), "lib.rs:292 open@gamma witness + shifted witness (g_2)"))
    return ex


def executed_ntt_bytes(H, K=None):
    return sum(64 << lg for lg, _, _ in ntt_executed(H, K))


def algorithmic_bytes(H, K=None):
    """NTT of size n moves 2*32*n bytes; MSM of size n moves n*(32+96)."""
    ntt_b = sum(64 << lg for lg, _, _ in ntt_inventory(H, K))
    big, small = msm_inventory(H, K)
    msm_b = sum(128 * n for n, _ in big)
    return ntt_b, msm_b
