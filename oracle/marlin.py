"""Restatement of Marlin<Fr, MarlinKZG10<Bls12_381>, SimpleHashFiatShamirRng<Blake2s, ChaChaRng>>
index / prove / verify (oracle only; canonical Python ints; small sizes).

Follows /root/reference src/lib.rs:79-148 (setup, index), 151-311 (prove), 315-433 (verify) and the
behaviour of ark-poly-commit 0.3 `kzg10` / `marlin_pc` (third-party, absent; SURVEY.md Appendix
B-3, B-4, B-6 [UPSTREAM-RECALLED]).  Pairing checks are replaced by the known-tau identity
(C - [v]G - [rv]gammaG == [tau - z]W), which is what e(.,.) verifies when tau is known.
"""
from .fields import R_MOD as R, Q_MOD, FQ_BYTES
from . import curve as EC
from . import ahp as AHP
from .poly import trim, poly_eval, divide_by_linear, degree
from .fs import SimpleHashFiatShamirRng, fr_rand, fr_bytes

PROTOCOL_NAME = b"MARLIN-2019"


# ----------------------------------------------------------------------------------
# SRS / keys (KZG10::setup + MarlinKZG10::trim with a known tau)
# ----------------------------------------------------------------------------------
class _GammaPowers:
    """powers_of_gamma_g[i] = [gamma tau^i]G on demand (KZG10::setup keeps max_degree + 2 of them; MarlinKZG10::trim
    keeps the first 3, SonicKZG10::trim additionally 3 per enforced degree bound at index max_degree - d + i)."""

    def __init__(self, tau, gamma_g):
        self.tau, self.gamma_g, self.cache = tau, gamma_g, {}

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[k] for k in range(*i.indices(1 << 62))]
        if i not in self.cache:
            self.cache[i] = EC.scalar_mul(self.gamma_g, pow(self.tau, i, R))
        return self.cache[i]


class SRS:
    def __init__(self, max_degree, tau, gamma, g=EC.G1_GEN):
        self.max_degree = max_degree
        self.tau, self.gamma = tau % R, gamma % R
        self.g = g
        self.powers_of_g = EC.srs_powers(tau, max_degree + 1, g)
        self.gamma_g = EC.scalar_mul(g, gamma)
        self.all_gamma = _GammaPowers(self.tau, self.gamma_g)
        self.powers_of_gamma_g = [self.all_gamma[i] for i in range(3)]   # hiding_bound + 2 = 3 kept by trim


def universal_setup(num_constraints, num_variables, num_non_zero, tau, gamma):
    return SRS(AHP.max_degree(num_constraints, num_variables, num_non_zero), tau, gamma)


def g1_bytes(pt):
    """ToBytes of GroupAffine: x || y (48 B LE canonical each) || infinity byte [B-6].
    The identity is (0, 1, true)."""
    if pt is None:
        return (0).to_bytes(FQ_BYTES, "little") + (1).to_bytes(FQ_BYTES, "little") + b"\x01"
    return pt[0].to_bytes(FQ_BYTES, "little") + pt[1].to_bytes(FQ_BYTES, "little") + b"\x00"


def commitment_bytes(c):
    """marlin_pc::Commitment::write: comm || shifted_exists || (shifted_comm or empty) [B-6];
    a sonic_pc commitment is a bare kzg10::Commitment = one G1Affine."""
    if c[1] == "sonic":
        return g1_bytes(c[0])
    comm, shifted = c
    return g1_bytes(comm) + (b"\x01" if shifted is not None else b"\x00") + g1_bytes(shifted[0] if shifted is not None else None)


def msm(srs, offset, coeffs):
    """KZG10::commit's MSM incl. skip_leading_zeros [B-3]; naive oracle arithmetic."""
    coeffs = list(coeffs)
    lz = 0
    while lz < len(coeffs) and coeffs[lz] % R == 0:
        lz += 1
    bases = srs.powers_of_g[offset + lz: offset + len(coeffs)]
    return EC.msm_pippenger(bases, coeffs[lz:]) if len(coeffs) > lz else None


def kzg_commit(srs, offset, poly, hiding_bound, rng):
    comm = msm(srs, offset, poly)
    blind = []
    if hiding_bound is not None:
        blind = trim([fr_rand(rng) for _ in range(hiding_bound + 2)])     # P::rand(hiding_bound + 1)
        comm = EC.add(comm, EC.msm_naive(srs.powers_of_gamma_g, blind))
    return comm, blind


def marlin_commit(srs, polys, rng):
    """MarlinKZG10::commit [B-4]: polys = [(label, coeffs, degree_bound, hiding_bound)].
    Returns commitments [(comm, (shifted,) or None)] and randomness [(rand, shifted_rand or None)]."""
    comms, rands = [], []
    for label, p, db, hb in polys:
        c, r = kzg_commit(srs, 0, p, hb, rng)
        if db is not None:
            sc, sr = kzg_commit(srs, srs.max_degree - db, p, hb, rng)
            comms.append((c, (sc,))); rands.append((r, sr))
        else:
            comms.append((c, None)); rands.append((r, None))
    return comms, rands


def sonic_commit(srs, polys, rng):
    """SonicKZG10::commit [SURVEY B-5, UPSTREAM-RECALLED]: ONE KZG10::commit per polynomial, against the shifted powers
    (powers_of_g[max_degree - d ..] and powers_of_gamma_g[max_degree - d + i]) when it has a degree bound d."""
    comms, rands = [], []
    for label, p, db, hb in polys:
        off = 0 if db is None else srs.max_degree - db
        comm = msm(srs, off, p)
        blind = []
        if hb is not None:
            blind = trim([fr_rand(rng) for _ in range(hb + 2)])
            comm = EC.add(comm, EC.msm_naive([srs.all_gamma[off + i] for i in range(len(blind))], blind))
        comms.append((comm, "sonic")); rands.append((blind, None))
    return comms, rands


def sonic_open(srs, polys, rands, point, xi):
    """SonicKZG10::open_individual_opening_challenges [B-5]: one combined polynomial (challenge xi^i for the i-th
    polynomial), one KZG10::open on the unshifted powers."""
    p, r = [], []
    for i, ((label, poly, db, hb), (rand, _)) in enumerate(zip(polys, rands)):
        ch = pow(xi, i, R)
        p = _axpy(p, ch, poly)
        r = _axpy(r, ch, rand)
    w = msm(srs, 0, divide_by_linear(p, point))
    random_v = None
    if trim(r):
        w = EC.add(w, EC.msm_naive(srs.powers_of_gamma_g, divide_by_linear(r, point)))
        random_v = poly_eval(r, point)
    return w, random_v


def pc_commit(pk_or_pc, srs, polys, rng):
    pc = pk_or_pc if isinstance(pk_or_pc, str) else pk_or_pc.pc
    return sonic_commit(srs, polys, rng) if pc == "sonic" else marlin_commit(srs, polys, rng)


def _axpy(acc, f, p):
    n = max(len(acc), len(p))
    out = [((acc[i] if i < len(acc) else 0) + f * (p[i] if i < len(p) else 0)) % R for i in range(n)]
    return trim(out)


def marlin_open(srs, enforced_bounds, polys, rands, point, xi):
    """MarlinKZG10::open_individual_opening_challenges + KZG10::open [B-4].
    polys: [(label, coeffs, degree_bound, hiding_bound)] in call order; rands aligned."""
    max_bound = max(enforced_bounds)
    p, r = [], []
    shifted_w, shifted_r, shifted_r_wit = [], [], []
    enforce = False
    ctr = 0
    for (label, poly, db, hb), (rand, srand) in zip(polys, rands):
        ch = pow(xi, ctr, R); ctr += 1
        p = _axpy(p, ch, poly)
        r = _axpy(r, ch, rand)
        if db is not None:
            enforce = True
            wit = divide_by_linear(poly, point)
            rwit = divide_by_linear(srand, point) if trim(srand) else None
            ch1 = pow(xi, ctr, R); ctr += 1
            sh = ([0] * (max_bound - db) + wit) if wit else []
            shifted_w = _axpy(shifted_w, ch1, sh)
            shifted_r = _axpy(shifted_r, ch1, srand)
            if rwit is not None:
                shifted_r_wit = _axpy(shifted_r_wit, ch1, rwit)
    # KZG10::open(powers, p, point, r)
    w = msm(srs, 0, divide_by_linear(p, point))
    random_v = None
    if trim(r):
        w = EC.add(w, EC.msm_naive(srs.powers_of_gamma_g, divide_by_linear(r, point)))
        random_v = poly_eval(r, point)
    if enforce:
        sw = msm(srs, srs.max_degree - max_bound, shifted_w)
        # open_with_witness_polynomial(shifted_powers, point, shifted_r, shifted_w, Some(shifted_r_witness))
        sw = EC.add(sw, EC.msm_naive(srs.powers_of_gamma_g, shifted_r_wit))
        srv = poly_eval(shifted_r, point)
        w = EC.add(w, sw)
        # marlin_pc `open`: `if let Some(s) = shifted_proof.random_v { random_v = random_v.map(|v| v + s) }` -- a None from
        # the unshifted proof (non-hiding combined polynomial) STAYS None; the shifted value is only ever added to a Some.
        # [ark-poly-commit 0.3 marlin_pc/mod.rs open_individual_opening_challenges, UPSTREAM-RECALLED; SURVEY.md B-4]
        random_v = (random_v + srv) % R if random_v is not None else None
    return w, random_v


# ----------------------------------------------------------------------------------
# Marlin::index / prove / verify
# ----------------------------------------------------------------------------------
class IndexKeys:
    pass


def marlin_index(srs, cs, pc="marlin"):
    """src/lib.rs:100-148: cs already padded/squared (AHP.pad_and_square).  pc: "marlin" = MarlinKZG10
    (src/test.rs:123), "sonic" = SonicKZG10 (benches/bench.rs:81)."""
    idx = AHP.index(cs)
    assert srs.max_degree >= idx.max_degree
    pk = IndexKeys()
    pk.pc = pc
    pk.index = idx
    pk.srs = srs
    pk.enforced_bounds = sorted([idx.domain_h.size - 2, idx.domain_k.size - 2])   # get_degree_bounds
    polys = [(l, idx.polys[l], None, None) for l in AHP.INDEXER_POLYNOMIALS]
    pk.index_polys = polys
    pk.index_comms, pk.index_rands = pc_commit(pc, srs, polys, None)
    return pk


def vk_bytes(pk):
    """IndexVerifierKey::write (data_structures.rs:36-43) + IndexInfo::write (indexer.rs:63-69)."""
    i = pk.index
    out = i.num_variables.to_bytes(8, "little") + i.num_constraints.to_bytes(8, "little") + i.num_non_zero.to_bytes(8, "little")
    for c in pk.index_comms:
        out += commitment_bytes(c)
    return out


class Proof:
    pass


def prove(pk, cs, zk_rng):
    """src/lib.rs:151-311."""
    idx, srs = pk.index, pk.srs
    st = AHP.prover_init(idx, cs)
    pub = AHP.public_input(st)
    fs = SimpleHashFiatShamirRng(PROTOCOL_NAME + vk_bytes(pk) + b"".join(fr_bytes(x) for x in pub))
    # round 1
    first = AHP.prover_first_round(st, zk_rng)
    c1, r1 = pc_commit(pk, srs, first, zk_rng)
    fs.absorb(b"".join(commitment_bytes(c) for c in c1))
    alpha, eta_a, eta_b, eta_c = AHP.verifier_first_round(st.domain_h, fs)
    # round 2
    second = AHP.prover_second_round(st, alpha, eta_a, eta_b, eta_c)
    c2, r2 = pc_commit(pk, srs, second, zk_rng)
    fs.absorb(b"".join(commitment_bytes(c) for c in c2))
    beta = AHP.verifier_second_round(st.domain_h, fs)
    # round 3
    third = AHP.prover_third_round(st, beta)
    c3, r3 = pc_commit(pk, srs, third, zk_rng)
    fs.absorb(b"".join(commitment_bytes(c) for c in c3))
    gamma = AHP.verifier_third_round(fs)

    polys = {l: (l, p, d, h) for l, p, d, h in pk.index_polys + first + second + third}
    rands = dict(zip([l for l, _, _, _ in pk.index_polys + first + second + third], pk.index_rands + r1 + r2 + r3))
    qs = AHP.query_set(beta, gamma)

    def ev(label, point):
        return poly_eval(polys[label][1], point)
    lcs = AHP.construct_linear_combinations(pub, ev, st.domain_h, st.domain_k, (alpha, eta_a, eta_b, eta_c, beta, gamma))

    def lc_eval(lc, point):
        return sum(c * (ev(t, point) if t is not None else 1) for c, t in lc) % R
    evaluations = []
    for label, _, point in qs:
        e = lc_eval(lcs[label], point)
        if label in ("inner_sumcheck", "outer_sumcheck"):
            assert e == 0, "sumcheck LC must vanish (src/ahp/mod.rs:177,214)"
        else:
            evaluations.append((label, e))
    evaluations = [e for _, e in sorted(evaluations)]
    fs.absorb(b"".join(fr_bytes(e) for e in evaluations))
    xi = fs.rand_u128_as_fr()

    # PC::open_combinations (lib.rs:292): build the LC polynomials, then batch_open per point
    lc_polys, lc_rands = {}, {}
    for label, lc in lcs.items():
        poly, rand, srand = [], [], None
        db, hb = None, None
        terms = [(c, t) for c, t in lc if t is not None]
        for c, t in terms:
            _, p, d, h = polys[t]
            if len(lc) == 1 and d is not None:
                assert c == 1
                db = d
            else:
                assert d is None
            if h is not None:
                hb = h if hb is None else max(hb, h)
            poly = _axpy(poly, c, p)
            rand = _axpy(rand, c, rands[t][0])
            if rands[t][1] is not None:
                srand = _axpy(srand or [], c, rands[t][1])
        lc_polys[label] = (label, poly, db, hb)
        lc_rands[label] = (rand, srand)
    proofs = []
    for pl, point in (("beta", beta), ("gamma", gamma)):
        labels = sorted(l for l, p, _ in qs if p == pl)
        if pk.pc == "sonic":
            proofs.append(sonic_open(srs, [lc_polys[l] for l in labels], [lc_rands[l] for l in labels], point, xi))
        else:
            proofs.append(marlin_open(srs, pk.enforced_bounds, [lc_polys[l] for l in labels], [lc_rands[l] for l in labels], point, xi))
    pr = Proof()
    pr.commitments = [c1, c2, c3]
    pr.evaluations = evaluations
    pr.pc_proof = proofs
    pr.challenges = dict(alpha=alpha, eta_a=eta_a, eta_b=eta_b, eta_c=eta_c, beta=beta, gamma=gamma, xi=xi)
    pr.polys = {l: polys[l][1] for l in polys}
    return pr


def proof_bytes(pr):
    """A flat, fully determined byte string of everything the proof contains (ToBytes layouts):
    9 commitments, 4 evaluations, per opening proof w || has_random_v || random_v."""
    out = b""
    for rnd in pr.commitments:
        for c in rnd:
            out += commitment_bytes(c)
    for e in pr.evaluations:
        out += fr_bytes(e)
    for w, rv in pr.pc_proof:
        out += g1_bytes(w) + (b"\x01" + fr_bytes(rv) if rv is not None else b"\x00" + bytes(32))
    return out


def verify(pk, public_input, pr, use_pairing=False):
    """src/lib.rs:315-433.  PC::check_combinations' KZG10 equation e(C - [v]G - [rv]gamma_G, H) == e(W, beta_H - [z]H)
    [ark-poly-commit 0.3 kzg10::check, UPSTREAM-RECALLED] is decided either with the real pairing of the curve
    (use_pairing=True: oracle/pairing.py, H = the G2 generator, beta_H = [tau]H -- what a verifier without tau does)
    or with the known-tau identity C - [v]G - [rv]gamma_G == [tau - z]W that the pairing equation
    is equivalent to (default: O(1) group operations, any curve, both PC schemes).  With use_pairing and SonicKZG10 the
    degree-bounded part of the combination is paired with [tau^-(max_degree - d)]H (sonic_pc `check_elems`
    [UPSTREAM-RECALLED, SURVEY B-5]) instead of being unshifted with tau."""
    idx, srs = pk.index, pk.srs
    pub = list(public_input)
    full = [1] + pub
    n = 1
    while n < len(full):
        n *= 2
    pub = pub + [0] * (n - len(full))                       # lib.rs:323-333
    fs = SimpleHashFiatShamirRng(PROTOCOL_NAME + vk_bytes(pk) + b"".join(fr_bytes(x) for x in pub))
    c1, c2, c3 = pr.commitments
    fs.absorb(b"".join(commitment_bytes(c) for c in c1))
    alpha, eta_a, eta_b, eta_c = AHP.verifier_first_round(idx.domain_h, fs)
    fs.absorb(b"".join(commitment_bytes(c) for c in c2))
    beta = AHP.verifier_second_round(idx.domain_h, fs)
    fs.absorb(b"".join(commitment_bytes(c) for c in c3))
    gamma = AHP.verifier_third_round(fs)
    fs.absorb(b"".join(fr_bytes(e) for e in pr.evaluations))
    xi = fs.rand_u128_as_fr()
    labels = AHP.INDEXER_POLYNOMIALS + ["w", "z_a", "z_b", "mask_poly", "t", "g_1", "h_1", "g_2", "h_2"]
    comms = dict(zip(labels, pk.index_comms + c1 + c2 + c3))
    bounds = {"g_1": idx.domain_h.size - 2, "g_2": idx.domain_k.size - 2}
    ev_map = dict(zip(["g_1", "g_2", "t", "z_b"], pr.evaluations))
    point_of = {"g_1": beta, "t": beta, "z_b": beta, "g_2": gamma}

    def ev(label, point):
        assert point_of[label] == point
        return ev_map[label]
    lcs = AHP.construct_linear_combinations(pub, ev, idx.domain_h, idx.domain_k, (alpha, eta_a, eta_b, eta_c, beta, gamma))
    qs = AHP.query_set(beta, gamma)
    G = srs.g
    ok = True
    for k, (pl, point) in enumerate((("beta", beta), ("gamma", gamma))):
        lbls = sorted(l for l, p, _ in qs if p == pl)
        combined, value = None, 0
        bounded_part, bounded_shift = None, None          # SonicKZG10 + pairing: the part committed against shifted powers
        ctr = 0
        for l in lbls:
            lc = lcs[l]
            # LC commitment and claimed evaluation (constant terms move to the evaluation side)
            const = sum(c for c, t in lc if t is None) % R
            claimed = (ev_map[l] if l in ev_map else 0)
            claimed = (claimed - const) % R
            lc_comm = None
            sonic = getattr(pk, "pc", "marlin") == "sonic"
            in_g2 = None
            for c, t in lc:
                if t is not None:
                    cm = comms[t][0]
                    if sonic and t in bounds:
                        # a degree-bounded Sonic commitment is [p(tau) tau^(max_degree - d)]G (hiding part shifted alike);
                        # the pairing check divides the shift out with the G2 element, the known-tau check with tau itself
                        if use_pairing:
                            assert len(lc) == 1
                            in_g2 = srs.max_degree - bounds[t]
                        else:
                            cm = EC.scalar_mul(cm, pow(srs.tau, -(srs.max_degree - bounds[t]), R))
                    lc_comm = EC.add(lc_comm, EC.scalar_mul(cm, c))
            ch = pow(xi, ctr, R); ctr += 1
            if in_g2 is not None:
                assert bounded_shift in (None, in_g2)
                bounded_part, bounded_shift = EC.add(bounded_part, EC.scalar_mul(lc_comm, ch)), in_g2
            else:
                combined = EC.add(combined, EC.scalar_mul(lc_comm, ch))
            value = (value + claimed * ch) % R
            if not sonic and len(lc) == 1 and lc[0][1] in bounds:
                t = lc[0][1]
                ch1 = pow(xi, ctr, R); ctr += 1
                shift_power = srs.powers_of_g[srs.max_degree - bounds[t]]
                adj = EC.add(comms[t][1][0], EC.neg(EC.scalar_mul(shift_power, claimed)))
                combined = EC.add(combined, EC.scalar_mul(adj, ch1))
        w, rv = pr.pc_proof[k]
        lhs = EC.add(combined, EC.neg(EC.scalar_mul(G, value)))
        if rv is not None:
            lhs = EC.add(lhs, EC.neg(EC.scalar_mul(srs.gamma_g, rv)))
        if use_pairing:
            from . import pairing as PR
            h = PR.G2_GEN
            beta_h = PR.g2_mul(h, srs.tau)                       # vk.beta_h of KZG10::setup
            inner = PR.g2_add(beta_h, PR.g2_neg(PR.g2_mul(h, point)))
            pairs = [(lhs, h), (EC.neg(w), inner)]
            if bounded_part is not None:                         # vk.degree_bounds_and_neg_powers_of_h of sonic_pc
                pairs.append((bounded_part, PR.g2_mul(h, pow(srs.tau, -bounded_shift, R))))
            ok = ok and PR.pairing_product_is_one(pairs)
        else:
            rhs = EC.scalar_mul(w, (srs.tau - point) % R)
            ok = ok and (lhs == rhs)
    return ok


# ----------------------------------------------------------------------------------
# Wire format: ark-serialize 0.3 `CanonicalSerialize` of `Proof` (src/data_structures.rs:100-110)
# ----------------------------------------------------------------------------------
# [UPSTREAM-RECALLED: ark-serialize / ark-ec / ark-ff 0.3, absent here]  derive(CanonicalSerialize) writes the fields in
# declaration order; Vec<T> = u64 LE length + items; Option<T> = one bool byte (+ the value); bool = one byte;
# Fp = ceil((MODULUS_BITS + flag bits) / 8) little-endian bytes of the canonical value with the flags in the top bits of the
# last byte; a short-Weierstrass affine point = x with SWFlags (bit 7: y > -y "positive", bit 6: infinity; the identity
# serialises x = 0); ProverMsg = Option<Vec<F>> (src/ahp/prover.rs:84-99); marlin_pc::Commitment = {comm, Option<shifted>};
# kzg10::Proof = {w, Option<random_v>}; BatchLCProof = {Vec<kzg10::Proof>, evals: Option<Vec<F>>} (None from Marlin).
def g1_compressed(pt):
    if pt is None:
        b = bytearray(FQ_BYTES); b[-1] |= 1 << 6
        return bytes(b)
    x, y = pt
    b = bytearray(x.to_bytes(FQ_BYTES, "little"))
    if y > (Q_MOD - y) % Q_MOD:
        b[-1] |= 1 << 7
    return bytes(b)


def g1_decompress(b):
    from .fields import G1_B as CURVE_B
    b = bytearray(b)
    flags = b[-1] & 0xC0
    b[-1] &= 0x3F
    x = int.from_bytes(b, "little")
    if flags & 0x40:
        assert x == 0 and not flags & 0x80, "invalid infinity encoding"
        return None
    assert x < Q_MOD
    y2 = (x * x * x + CURVE_B) % Q_MOD
    y = pow(y2, (Q_MOD + 1) // 4, Q_MOD)                 # p = 3 mod 4 for both curves
    assert y * y % Q_MOD == y2, "x is not on the curve"
    if (y > (Q_MOD - y) % Q_MOD) != bool(flags & 0x80):
        y = (Q_MOD - y) % Q_MOD
    return (x, y)


def _u64(n):
    return int(n).to_bytes(8, "little")


def proof_serialize(pr):
    """CanonicalSerialize bytes of `Proof<Fr, MarlinKZG10 | SonicKZG10>` (855 bytes for MarlinKZG10 on BLS12-381 with
    random_v = Some at beta and None at gamma; the README's 880 counts 13 G1 + 8 Fr of an older layout)."""
    out = _u64(len(pr.commitments))
    for rnd in pr.commitments:
        out += _u64(len(rnd))
        for c in rnd:
            if c[1] == "sonic":
                out += g1_compressed(c[0])
            else:
                out += g1_compressed(c[0]) + (b"\x01" + g1_compressed(c[1][0]) if c[1] is not None else b"\x00")
    out += _u64(len(pr.evaluations)) + b"".join(fr_bytes(e) for e in pr.evaluations)
    out += _u64(3) + b"\x00" * 3                        # prover_messages: three EmptyMessage = Option::None each
    out += _u64(len(pr.pc_proof))
    for w, rv in pr.pc_proof:
        out += g1_compressed(w) + (b"\x01" + fr_bytes(rv) if rv is not None else b"\x00")
    out += b"\x00"                                      # BatchLCProof.evals = None
    return out


def proof_deserialize(b, pc="marlin"):
    """inverse of proof_serialize (CanonicalDeserialize with point validation: on-curve + flag consistency)."""
    pos = [0]

    def take(n):
        assert pos[0] + n <= len(b), "truncated proof"
        v = b[pos[0]:pos[0] + n]; pos[0] += n
        return v

    def u64(): return int.from_bytes(take(8), "little")

    def boolean():
        v = take(1)[0]
        assert v in (0, 1)
        return bool(v)

    def fr():
        v = int.from_bytes(take(32), "little")
        assert v < R
        return v
    pr = Proof()
    pr.commitments = []
    for _ in range(u64()):
        rnd = []
        for _ in range(u64()):
            comm = g1_decompress(take(FQ_BYTES))
            if pc == "sonic":
                rnd.append((comm, "sonic"))
            else:
                rnd.append((comm, (g1_decompress(take(FQ_BYTES)),) if boolean() else None))
        pr.commitments.append(rnd)
    pr.evaluations = [fr() for _ in range(u64())]
    for _ in range(u64()):
        assert not boolean(), "Marlin's prover messages are all EmptyMessage"
    pr.pc_proof = []
    for _ in range(u64()):
        w = g1_decompress(take(FQ_BYTES))
        pr.pc_proof.append((w, fr() if boolean() else None))
    assert not boolean() and pos[0] == len(b)
    return pr
