"""CPU recomputation of the polynomial-commitment layer of one proof at sizes the pure-Python oracle cannot reach
(test infrastructure, oracle side).

Input: the prover's polynomials as the device exported them (`mh_marlin_get_poly`), the flat proof, the SRS.  Output:
every commitment and both opening proofs (W, random_v) recomputed the way the REFERENCE computes them -- the call
sequence of ark-poly-commit 0.3 `marlin_pc` / `sonic_pc` / `kzg10` (third-party, absent; SURVEY.md Appendix B-3..B-5
[UPSTREAM-RECALLED]) under /root/reference src/lib.rs:172,193,213 (PC::commit) and 292-302 (PC::open_combinations):
one MSM per KZG10::commit, a separate MSM for each shifted commitment / shifted witness, results added as group
elements.  None of the product's reformulations (merged MSMs, fused divisions, shared sorts) is repeated here; the big
vector arithmetic runs in the C restatement (oracle/cref.py), the small and the protocol logic in oracle/*.py.

This mirrors oracle/marlin.py `prove` from the Fiat-Shamir replay on (the challenges come from the proof's own
commitment bytes, like the verifier's), so together with the byte-identical proofs at <= 2^14 it pins the whole proof
at 2^18 .. 2^22: commitments and openings are unique group elements given the polynomials, and the polynomials that
are not re-derived here (the AHP rounds) are exactly what the verifier's sumcheck identities constrain.
"""
import numpy as np
from oracle import fields as F, curve as EC, ahp as AHP, marlin as MR, cref
from oracle.poly import Domain, poly_eval, divide_by_linear, trim
from oracle.fs import SimpleHashFiatShamirRng, fr_bytes
from tests.util import limbs_to_fq
from tests.verify_adapter import parse_proof, _comm_len

R = F.R_MOD
PROVER_LABELS = ["w", "z_a", "z_b", "mask_poly", "t", "g_1", "h_1", "g_2", "h_2"]


def _pt(xyz):
    xy, inf = cref.g1_to_affine(xyz)
    L = F.FQ_LIMBS64
    return None if inf else (limbs_to_fq(xy[:L]), limbs_to_fq(xy[L:]))


class CpuPC:
    def __init__(self, bases, max_degree, tau, gamma, threads):
        """bases: (max_degree + 1, 2 * FQL) uint64 = powers_of_g as downloaded from the device; tau / gamma only serve the
        3-coefficient hiding parts (powers_of_gamma_g[i] = [gamma tau^i]G, a handful of scalar multiplications)."""
        self.bases, self.D, self.tau, self.gamma, self.threads = bases, max_degree, tau % R, gamma % R, threads
        self._gp = {}

    def gamma_power(self, i):
        if i not in self._gp:
            self._gp[i] = EC.scalar_mul(EC.G1_GEN, self.gamma * pow(self.tau, i, R) % R)
        return self._gp[i]

    def msm(self, offset, coeffs):
        """KZG10::commit's MSM over powers_of_g[offset ..] (skip_leading_zeros changes nothing in the sum)."""
        coeffs = np.ascontiguousarray(coeffs, dtype=np.uint64)
        if len(coeffs) == 0:
            return None
        assert offset + len(coeffs) <= len(self.bases)
        return _pt(cref.msm(self.bases[offset:offset + len(coeffs)], coeffs, montgomery=True, threads=self.threads))

    def hiding(self, offset, blind):
        blind = trim(list(blind))
        return EC.msm_naive([self.gamma_power(offset + i) for i in range(len(blind))], blind) if blind else None


def replay_challenges(vk_bytes, pub, flat, pc, H, K):
    """Marlin::verify's transcript replay (src/lib.rs:335-360) on the flat proof: the commitment and evaluation bytes of
    the flat layout ARE the absorbed bytes."""
    cl = _comm_len(pc)
    fs = SimpleHashFiatShamirRng(MR.PROTOCOL_NAME + vk_bytes + b"".join(fr_bytes(x) for x in pub))
    dh, dk = Domain(H), Domain(K)
    fs.absorb(flat[0:4 * cl])
    alpha, eta_a, eta_b, eta_c = AHP.verifier_first_round(dh, fs)
    fs.absorb(flat[4 * cl:7 * cl])
    beta = AHP.verifier_second_round(dh, fs)
    fs.absorb(flat[7 * cl:9 * cl])
    gamma = AHP.verifier_third_round(fs)
    fs.absorb(flat[9 * cl:9 * cl + 128])
    xi = fs.rand_u128_as_fr()
    return dict(alpha=alpha, eta_a=eta_a, eta_b=eta_b, eta_c=eta_c, beta=beta, gamma=gamma, xi=xi), dh, dk


def recompute_commitments(cpc, polys, zk, H, K, pc):
    """PC::commit of the three rounds (lib.rs:172,193,213).  zk: tests/zkstream.prove_zk_draws.  Returns
    {label: commitment in the oracle's form} and {label: (rand, shifted_rand or None)}."""
    bounds = {"g_1": H - 2, "g_2": K - 2}
    # rng order = call order: w, z_a, z_b hiding; g_1 hiding, then its shifted commitment draws again (MarlinKZG10 only)
    blind = {"w": zk["blind_w"], "z_a": zk["blind_za"], "z_b": zk["blind_zb"], "g_1": zk["blind_g1"]}
    comms, rands = {}, {}
    for l in PROVER_LABELS:
        p = polys[l]
        b = blind.get(l)
        if pc == "sonic":
            off = cpc.D - bounds[l] if l in bounds else 0
            c = cpc.msm(off, p)
            if b is not None:
                c = EC.add(c, cpc.hiding(off, b))
            comms[l] = (c, "sonic")
            rands[l] = (trim(list(b)) if b is not None else [], None)
            continue
        c = cpc.msm(0, p)
        if b is not None:
            c = EC.add(c, cpc.hiding(0, b))
        if l in bounds:
            sb = zk["blind_g1_shifted"] if l == "g_1" else None
            sc = cpc.msm(cpc.D - bounds[l], p)
            if sb is not None:
                sc = EC.add(sc, cpc.hiding(0, sb))
            comms[l] = (c, (sc,))
            rands[l] = (trim(list(b)) if b is not None else [], trim(list(sb)) if sb is not None else [])
        else:
            comms[l] = (c, None)
            rands[l] = (trim(list(b)) if b is not None else [], None)
    return comms, rands


def _lc_polys(lcs, polys, rands, bounds, hiding, threads):
    """open_combinations' LC polynomials and randomness (marlin_pc / sonic_pc `open_combinations`; oracle/marlin.py
    prove does the same on Python lists)."""
    out_p, out_r = {}, {}
    for label, lc in lcs.items():
        terms = [(c, t) for c, t in lc if t is not None]
        db, rand, srand = None, [], None
        if len(lc) == 1 and terms[0][1] in bounds:
            assert terms[0][0] == 1
            db = bounds[terms[0][1]]
        for c, t in terms:
            assert db is not None or t not in bounds
            rand = MR._axpy(rand, c, rands[t][0] if t in rands else [])
            if t in rands and rands[t][1] is not None:
                srand = MR._axpy(srand or [], c, rands[t][1])
        if len(terms) == 1 and terms[0][0] == 1:
            poly = polys[terms[0][1]]
        else:
            poly = cref.lincomb([(c, polys[t]) for c, t in terms], threads=threads)
        out_p[label] = (poly, db, any(t in hiding for _, t in terms))
        out_r[label] = (rand, srand)
    return out_p, out_r


def marlin_open(cpc, enforced_bounds, items, point, xi):
    """MarlinKZG10::open (oracle/marlin.py marlin_open on numpy vectors): items = [(poly, degree_bound, rand, srand)]."""
    max_bound = max(enforced_bounds)
    terms, r = [], []
    shifted_w = None
    shifted_r, shifted_r_wit = [], []
    enforce = False
    ctr = 0
    for poly, db, rand, srand in items:
        ch = pow(xi, ctr, R); ctr += 1
        terms.append((ch, poly))
        r = MR._axpy(r, ch, rand)
        if db is not None:
            enforce = True
            wit, _ = cref.div_linear(poly, point)
            ch1 = pow(xi, ctr, R); ctr += 1
            pad = max_bound - db
            if shifted_w is None:
                shifted_w = np.zeros((0, 4), dtype=np.uint64)
            need = pad + len(wit)
            if need > len(shifted_w):
                shifted_w = np.concatenate([shifted_w, np.zeros((need - len(shifted_w), 4), dtype=np.uint64)])
            cref.add_at(shifted_w, pad, cref.lincomb([(ch1, wit)]))
            shifted_r = MR._axpy(shifted_r, ch1, srand or [])
            if trim(list(srand or [])):
                shifted_r_wit = MR._axpy(shifted_r_wit, ch1, divide_by_linear(srand, point))
    p = cref.lincomb(terms, threads=cpc.threads)
    wit, _ = cref.div_linear(p, point)
    w = cpc.msm(0, wit)
    random_v = None
    if trim(list(r)):
        w = EC.add(w, cpc.hiding(0, divide_by_linear(r, point)))
        random_v = poly_eval(r, point)
    if enforce:
        sw = cpc.msm(cpc.D - max_bound, shifted_w)
        sw = EC.add(sw, cpc.hiding(0, shifted_r_wit))
        w = EC.add(w, sw)
        # `random_v.map(|v| v + s)`: a None stays None (oracle/marlin.py marlin_open)
        random_v = (random_v + poly_eval(shifted_r, point)) % R if random_v is not None else None
    return w, random_v


def sonic_open(cpc, items, point, xi):
    """SonicKZG10::open: one combined polynomial, one KZG10::open on the unshifted powers."""
    terms, r = [], []
    for i, (poly, db, rand, srand) in enumerate(items):
        ch = pow(xi, i, R)
        terms.append((ch, poly))
        r = MR._axpy(r, ch, rand)
    p = cref.lincomb(terms, threads=cpc.threads)
    wit, _ = cref.div_linear(p, point)
    w = cpc.msm(0, wit)
    random_v = None
    if trim(list(r)):
        w = EC.add(w, cpc.hiding(0, divide_by_linear(r, point)))
        random_v = poly_eval(r, point)
    return w, random_v


def recompute_proof(cpc, polys, zk, vk_bytes, pub, flat, H, K, pc="marlin"):
    """Everything of the proof that is a function of the polynomials: 9 commitments, 4 evaluations, 2 openings.
    polys: label -> (len,4) uint64 Montgomery for the 9 prover and 6 indexer polynomials.  Returns an oracle Proof."""
    ch, dh, dk = replay_challenges(vk_bytes, pub, flat, pc, H, K)
    beta, gamma, xi = ch["beta"], ch["gamma"], ch["xi"]
    comms, rands = recompute_commitments(cpc, polys, zk, H, K, pc)
    point_of = {"g_1": beta, "t": beta, "z_b": beta, "g_2": gamma}
    evals = {l: cref.poly_eval(polys[l], z) for l, z in point_of.items()}
    lcs = AHP.construct_linear_combinations(pub, lambda l, z: evals[l] if point_of[l] == z else None, dh, dk,
                                            (ch["alpha"], ch["eta_a"], ch["eta_b"], ch["eta_c"], beta, gamma))
    bounds = {"g_1": H - 2, "g_2": K - 2}
    lc_p, lc_r = _lc_polys(lcs, polys, rands, bounds, {"w", "z_a", "z_b", "g_1"}, cpc.threads)
    # the sumcheck LCs must vanish at their points (src/ahp/mod.rs:177,214): constants included
    for l, z in (("outer_sumcheck", beta), ("inner_sumcheck", gamma)):
        const = sum(c for c, t in lcs[l] if t is None) % R
        assert (cref.poly_eval(lc_p[l][0], z) + const) % R == 0, l
    qs = AHP.query_set(beta, gamma)
    proofs = []
    for pl, point in (("beta", beta), ("gamma", gamma)):
        labels = sorted(l for l, p, _ in qs if p == pl)
        items = [(lc_p[l][0], lc_p[l][1], lc_r[l][0], lc_r[l][1]) for l in labels]
        proofs.append(sonic_open(cpc, items, point, xi) if pc == "sonic" else marlin_open(cpc, sorted(bounds.values()), items, point, xi))
    pr = MR.Proof()
    pr.commitments = [[comms[l] for l in PROVER_LABELS[0:4]], [comms[l] for l in PROVER_LABELS[4:7]], [comms[l] for l in PROVER_LABELS[7:9]]]
    pr.evaluations = [evals[l] for l in sorted(evals)]
    pr.pc_proof = proofs
    return pr
