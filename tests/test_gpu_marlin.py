"""End-to-end parity of the device-resident prover (mh_marlin_index / mh_marlin_prove through the
C ABI) against the oracle: byte-identical proofs on the reference's own test shapes (src/test.rs:
165-203) and benchmark circuit (benches/bench.rs), checked (a) against the committed golden
fixtures and (b) against a fresh oracle run, and verified with the oracle's verifier
(accepts; rejects a wrong public input, src/test.rs:158-161)."""
import json
import os
import pytest
from oracle import ahp as AHP, marlin as MR, fs as FS, fields as F
from marlin_amd import marlin as GM

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "marlin_proofs.json")))
TAU, GAMMA, SEED = int(GOLD["tau"], 16), int(GOLD["gamma"], 16), bytes.fromhex(GOLD["zk_seed"])


def _gpu_prove(case):
    nc, nv = case["num_constraints"], case["num_variables"]
    a, b = int(case["a"], 16), int(case["b"], 16)
    srs = GM.universal_setup(max(nc, nv), max(nc, nv), 3 * max(nc, nv), TAU, GAMMA)
    assert srs.max_degree == case["srs_max_degree"]
    build = GM.test_circuit if case["kind"] == "test_circuit" else GM.dummy_circuit
    if case["kind"] == "test_circuit":
        ncp, ni, mats, inst, wit = build(a, b, nc, nv)
    else:
        ncp, ni, mats, inst, wit = build(a, b, nv, nc)
    pk = GM.index(srs, ncp, ni, mats)
    assert (pk.H, pk.K) == (case["H"], case["K"])
    return pk, GM.prove(pk, inst, wit, SEED)


@pytest.mark.skipif(F.CURVE != "bls12_381", reason="golden fixtures are BLS12-381 / MarlinKZG10")
@pytest.mark.parametrize("case", GOLD["cases"], ids=lambda c: "%s-%d-%d" % (c["kind"], c["num_constraints"], c["num_variables"]))
def test_proof_bytes_match_golden(gpu, case):
    import hashlib
    pk, proof = _gpu_prove(case)
    assert hashlib.blake2s(pk.vk_bytes()).hexdigest() == case["vk_bytes_blake2s"]     # index commitments
    assert len(proof) == GM.PROOF_BYTES == GM.proof_bytes_len("marlin")
    assert proof.hex() == case["proof_bytes"]


def test_prover_polynomials_match_oracle(gpu):
    """Every prover oracle polynomial (coefficient form) equals the oracle's, for a test.rs shape."""
    from tests.util import np_to_fr
    from oracle.poly import trim
    rng = FS.test_rng()
    a, b = FS.fr_rand(rng), FS.fr_rand(rng)
    nc, nv = 25, 25
    cs = AHP.pad_and_square(AHP.finalize_test_circuit(AHP.test_circuit(a, b, nc, nv)))
    srs_o = MR.universal_setup(nc, nv, 3 * nc, TAU, GAMMA)
    pk_o = MR.marlin_index(srs_o, cs)
    pr = MR.prove(pk_o, cs, FS.ChaChaRng(SEED, 20))
    srs = GM.universal_setup(nc, nv, 3 * nc, TAU, GAMMA)
    ncp, ni, mats, inst, wit = GM.test_circuit(a, b, nc, nv)
    pk = GM.index(srs, ncp, ni, mats)
    GM.prove(pk, inst, wit, SEED)
    bad = []
    for label in ["row", "col", "a_val", "b_val", "c_val", "row_col", "w", "z_a", "z_b", "mask_poly", "t", "g_1", "h_1", "g_2", "h_2"]:
        got = trim(np_to_fr(pk.get_poly(label)))
        if got != trim(pr.polys[label]):
            bad.append(label)
    assert not bad, bad


def test_proof_matches_fresh_oracle_and_verifies(gpu):
    """A size not in the fixtures: DummyCircuit with 2^7 constraints; oracle run here."""
    rng = FS.test_rng()
    a, b = FS.fr_rand(rng), FS.fr_rand(rng)
    nc, nv = 128, 10
    cs = AHP.pad_and_square(AHP.dummy_circuit(a, b, nv, nc))
    srs_o = MR.universal_setup(nc, nc, 3 * nc, TAU, GAMMA)
    pk_o = MR.marlin_index(srs_o, cs)
    pr = MR.prove(pk_o, cs, FS.ChaChaRng(SEED, 20))
    srs = GM.universal_setup(nc, nc, 3 * nc, TAU, GAMMA)
    ncp, ni, mats, inst, wit = GM.dummy_circuit(a, b, nv, nc)
    pk = GM.index(srs, ncp, ni, mats)
    assert pk.vk_bytes() == MR.vk_bytes(pk_o)
    proof = GM.prove(pk, inst, wit, SEED)
    assert proof == MR.proof_bytes(pr)
    assert MR.verify(pk_o, [a * b % F.R_MOD], pr) and not MR.verify(pk_o, [a], pr)
    # a different zk seed gives a different (still valid-looking) proof: the rng really is consumed
    assert GM.prove(pk, inst, wit, bytes(32)) != proof


@pytest.mark.parametrize("log_n", [12, 16, 18, 20, 22] if F.CURVE == "bls12_381" else [12, 16, 20])
def test_full_size_proof_verifies(gpu, log_n):
    """BASELINE.json sizes (2^18 = configs[1], 2^20 = configs[2]): the proof of DummyCircuit made on the
    device verifies under the oracle's verifier, and a wrong public input / tampered proof is rejected
    (the reference's own acceptance property, src/test.rs:158-161)."""
    from tests.verify_adapter import oracle_verify
    rng = FS.test_rng()
    a, b = FS.fr_rand(rng), FS.fr_rand(rng)
    n = 1 << log_n
    srs = GM.universal_setup(n, n, 3 * n, TAU, GAMMA)
    ncp, ni, mats, inst, wit = GM.dummy_circuit(a, b, 10, n)
    pk = GM.index(srs, ncp, ni, mats)
    assert (pk.H, pk.K) == (n, 4 * n)
    # the index built the fixed-base window table for powers_of_g (automatic width) ...
    from tests.util import auto_window_bits
    assert srs.powers_of_g.table_info()[0] == auto_window_bits(srs.max_degree + 1)
    fb0, vb0 = gpu.msm_path_counts()
    proof = GM.prove(pk, inst, wit, SEED)
    # ... and every MSM batch of the proof ran on it: three commit rounds and the openings, no variable-base group
    assert gpu.msm_path_counts() == (fb0 + 4, vb0)
    vk = pk.vk_bytes()
    c = a * b % F.R_MOD
    assert oracle_verify(vk, srs.max_degree, TAU, GAMMA, [c], proof)
    assert not oracle_verify(vk, srs.max_degree, TAU, GAMMA, [a], proof)
    bad = bytearray(proof); bad[len(proof) - 2 * (2 * F.FQ_BYTES + 34) - 128 + 3] ^= 1     # flip a bit of an evaluation
    assert not oracle_verify(vk, srs.max_degree, TAU, GAMMA, [c], bytes(bad))
    assert GM.prove(pk, inst, wit, SEED) == proof                    # deterministic given the zk seed


@pytest.mark.parametrize("log_n,pc", [(10, "marlin"), (16, "marlin"), (12, "sonic")])
def test_device_proof_passes_the_products_own_verifier(gpu, log_n, pc):
    """prove on the device -> Marlin::verify on the host with the real pairing (mh_marlin_verify), no tau on the verifier's
    side, both PC schemes, on the curve of this process: accepts; a wrong public input and a tampered evaluation are
    rejected (src/test.rs:158-161).  The oracle only supplies a G2 point for h (any point of G2 will do: kzg10::setup
    draws it at random)."""
    from oracle import g2 as G2
    from tests.util import fr_to_np, fq_to_limbs
    import numpy as np
    rng = FS.test_rng()
    a, b = FS.fr_rand(rng), FS.fr_rand(rng)
    n = 1 << log_n
    srs = GM.universal_setup(n, n, 3 * n, TAU, GAMMA, pc=pc)
    ncp, ni, mats, inst, wit = GM.dummy_circuit(a, b, 10, n)
    pk = GM.index(srs, ncp, ni, mats, pc=pc)
    proof = GM.prove(pk, inst, wit, SEED)
    h = G2.g2_mul(G2.G2_GEN, 0x5eed)
    h_np = np.array(sum([fq_to_limbs(c) for c in (h[0][0], h[0][1], h[1][0], h[1][1])], []), dtype=np.uint64)
    els = srs.verifier_key(pk, h_np, pc=pc)
    c = a * b % F.R_MOD
    assert GM.verify(pk.vk_bytes(), *els, fr_to_np([c]), proof, pc=pc)
    assert not GM.verify(pk.vk_bytes(), *els, fr_to_np([a]), proof, pc=pc)
    # through the wire: CanonicalSerialize bytes (compressed points) -> validating deserializer -> verifier
    wire = GM.proof_serialize(proof, pc)
    assert len(wire) < len(proof) and GM.proof_deserialize(wire, pc) == proof
    assert GM.verify(pk.vk_bytes(), *els, fr_to_np([c]), GM.proof_deserialize(wire, pc), pc=pc)
    bad = bytearray(proof); bad[len(proof) - 2 * (2 * F.FQ_BYTES + 34) - 128 + 3] ^= 1
    assert not GM.verify(pk.vk_bytes(), *els, fr_to_np([c]), bytes(bad), pc=pc)


@pytest.mark.parametrize("pc", ["marlin", "sonic"])
def test_prove_and_verify_with_the_callers_fiat_shamir(gpu, pc):
    """Marlin<F, PC, FS> is generic over FS: FiatShamirRng (src/lib.rs:64-70; VERDICT r03 missing 5).  mh_marlin_prove_fs routes
    initialize / absorb / next_u64 to the caller: (i) with the oracle's SimpleHashFiatShamirRng<Blake2s, ChaChaRng> behind the
    callbacks the proof is the built-in entry point's proof, byte for byte, and the transcript the callbacks saw is the
    reference's (initialize once with "MARLIN-2019" || vk || input, then one absorb per round and one for the evaluations);
    (ii) with ANOTHER FS (SHA-256 in counter mode) the proof differs, verifies under mh_marlin_verify_fs with that FS, and is
    rejected by the built-in verifier and for a wrong input."""
    import hashlib
    import numpy as np
    from oracle import g2 as G2
    from tests.util import fr_to_np, fq_to_limbs
    rng = FS.test_rng()
    a, b = FS.fr_rand(rng), FS.fr_rand(rng)
    n = 1 << 10
    srs = GM.universal_setup(n, n, 3 * n, TAU, GAMMA, pc=pc)
    ncp, ni, mats, inst, wit = GM.dummy_circuit(a, b, 10, n)
    pk = GM.index(srs, ncp, ni, mats, pc=pc)
    want = GM.prove(pk, inst, wit, SEED)

    class Ref:                                   # the reference's instantiation, through the callbacks
        def __init__(self):
            self.log, self.fs = [], None
        def initialize(self, data):
            self.log.append(("initialize", len(data))); self.fs = FS.SimpleHashFiatShamirRng(data); self.first = data
        def absorb(self, data):
            self.log.append(("absorb", len(data))); self.fs.absorb(data)
        def next_u64(self):
            return self.fs.next_u64()
    ref = Ref()
    assert GM.prove_fs(pk, inst, wit, SEED, ref) == want
    assert [k for k, _ in ref.log] == ["initialize", "absorb", "absorb", "absorb", "absorb"]
    assert ref.first.startswith(b"MARLIN-2019" + pk.vk_bytes())

    class Sha:                                   # some other FiatShamirRng: SHA-256 state, counter-mode output
        def initialize(self, data):
            self.state, self.ctr = hashlib.sha256(b"init" + data).digest(), 0
        def absorb(self, data):
            self.state, self.ctr = hashlib.sha256(self.state + data).digest(), 0
        def next_u64(self):
            self.ctr += 1
            return int.from_bytes(hashlib.sha256(self.state + self.ctr.to_bytes(8, "little")).digest()[:8], "little")
    other = GM.prove_fs(pk, inst, wit, SEED, Sha())
    assert other != want and len(other) == len(want)
    h = G2.g2_mul(G2.G2_GEN, 0x5eed)
    h_np = np.array(sum([fq_to_limbs(c) for c in (h[0][0], h[0][1], h[1][0], h[1][1])], []), dtype=np.uint64)
    els = srs.verifier_key(pk, h_np, pc=pc)
    c = a * b % F.R_MOD
    assert GM.verify_fs(pk.vk_bytes(), *els, fr_to_np([c]), other, Sha(), pc=pc)
    assert not GM.verify_fs(pk.vk_bytes(), *els, fr_to_np([a]), other, Sha(), pc=pc)
    assert not GM.verify(pk.vk_bytes(), *els, fr_to_np([c]), other, pc=pc)              # another transcript, other challenges
    assert GM.verify_fs(pk.vk_bytes(), *els, fr_to_np([c]), want, Ref(), pc=pc)          # and the built-in proof under the callbacks

    # an exception inside the caller's FS must come out of prove_fs / verify_fs -- not be printed by ctypes and
    # swallowed, with the library carrying on over a transcript of zeros
    class Boom(Sha):
        def __init__(self, fail_at):
            self.n, self.fail_at = 0, fail_at
        def absorb(self, data):
            self.n += 1
            if self.n == self.fail_at:
                raise KeyError("transcript broke at absorb %d" % self.n)
            Sha.absorb(self, data)
    with pytest.raises(KeyError, match="absorb 2"):
        GM.prove_fs(pk, inst, wit, SEED, Boom(2))
    with pytest.raises(KeyError, match="absorb 1"):
        GM.verify_fs(pk.vk_bytes(), *els, fr_to_np([c]), other, Boom(1), pc=pc)
    assert GM.prove_fs(pk, inst, wit, SEED, Sha()) == other                               # the library is usable afterwards


TOGGLE_WORKER = r'''
import sys
sys.path.insert(0, %(root)r)
import marlin_amd as M
from marlin_amd import marlin as GM
M.init(0)
n = 1 << %(log_n)d
srs = GM.universal_setup(n, n, 3 * n, %(tau)d, %(gamma)d, pc=%(pc)r)
nc, ni, mats, inst, wit = GM.dummy_circuit(%(a)d, %(b)d, 10, n)
pk = GM.index(srs, nc, ni, mats, pc=%(pc)r)
sys.stdout.write(pk.vk_bytes().hex() + " " + GM.prove(pk, inst, wit, bytes(range(32))).hex())
'''


@pytest.mark.parametrize("env,pc", [({"MH_FB": "0"}, "marlin"), ({"MH_FB": "0"}, "sonic"), ({"MH_NTT": "32"}, "marlin"), ({"MH_ACC_PARTS": "0"}, "marlin")],
                         ids=lambda v: v if isinstance(v, str) else ",".join("%s=%s" % kv for kv in v.items()))
def test_alternative_paths_give_the_same_bytes(gpu, env, pc):
    """The cross-check paths the library keeps behind switches -- variable-base MSM for every commitment (what serves a key
    whose window table does not fit), the 32-bit-limb NTT kernel and the one-thread-per-bucket accumulate kernel of rounds 1-5
    (MH_ACC_PARTS=0) -- produce the same index commitments and the same proof, byte
    for byte, as the default path (fixed-base MSM over virtual slots, 30-bit NTT with Shoup twiddle products).  (2^13 constraints: the bucket sets
    have empty buckets.)  The tuning switches of rounds 3-4 (two-stream pipeline, quad-lane reduction, resident-wave override,
    unshared sorts, ...) are gone with the paths they selected."""
    import subprocess, sys
    a, b, log_n = 0x1234567, 0x7654321, 13
    n = 1 << log_n
    srs = GM.universal_setup(n, n, 3 * n, TAU, GAMMA, pc=pc)
    ncp, ni, mats, inst, wit = GM.dummy_circuit(a, b, 10, n)
    pk = GM.index(srs, ncp, ni, mats, pc=pc)
    want = pk.vk_bytes().hex() + " " + GM.prove(pk, inst, wit, bytes(range(32))).hex()
    code = TOGGLE_WORKER % dict(root=ROOT, log_n=log_n, tau=TAU, gamma=GAMMA, a=a, b=b, pc=pc)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.strip() == want


SHARD_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import torch.distributed as dist
import marlin_amd as M
from marlin_amd import marlin as GM, dist as MD
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
M.init(0)                                   # the GPU box has one GPU: both ranks share it
n = 1 << %(log_n)d
srs = GM.universal_setup(n, n, 3 * n, %(tau)d, %(gamma)d, pc=%(pc)r)
nc, ni, mats, inst, wit = GM.dummy_circuit(%(a)d, %(b)d, 10, n)
pk = GM.index(srs, nc, ni, mats, pc=%(pc)r)
MD.enable_sharded_prove(dist)
if %(sliced)d:
    # rounds 2 and 3 run on slices (distributed transforms, one all-gather per round); sliced = 2: the round polynomials go
    # through the all-to-all (every rank sending `world` copies) instead of the device all-gather
    MD.enable_alltoall(dist, allgather_dev=%(sliced)d == 1)
proof = GM.prove(pk, inst, wit, bytes(range(32)))
proof2 = GM.prove(pk, inst, wit, bytes(range(1, 33)))        # the key's sliced tables are reused
open(os.path.join(%(out)r, "proof%%d.bin" %% rank), "wb").write(proof + proof2)
dist.barrier(); dist.destroy_process_group()
'''


# (the native transport runs the same matrix over its stand-in in tests/test_gpu_rccl_native.py; the callback transport keeps one case
# per distinct situation here: 2 / 3 / 4 / 8 ranks, fewer partitions than ranks, both PC schemes, sliced with and without the device
# all-gather, a rank count that is not a power of two)
@pytest.mark.parametrize("world,log_n,pc,sliced", [(2, 12, "marlin", 0), (3, 12, "marlin", 0), (2, 16, "sonic", 0), (8, 12, "marlin", 0),
                                                   (8, 16, "marlin", 0), (2, 12, "marlin", 1), (4, 12, "sonic", 1), (4, 16, "marlin", 1),
                                                   (8, 14, "sonic", 1), (3, 12, "marlin", 1), (4, 12, "marlin", 2)])
def test_sharded_prove_ranks_equal_single(gpu, tmp_path, world, log_n, pc, sliced):
    """MSM sharding by bucket range across 2, 3, 4 and 8 ranks (gloo exchange, all ranks on the one GPU of this box), both PC
    schemes, yields the very same proof bytes as the unsharded prover.  At 2^12 the window table has 2 partitions (c = 13:
    2^12 buckets), fewer than 3, 4 or 8 ranks: those groups run unsharded on every rank and rank 0's copy counts; at 2^16
    it has 64 and every rank owns 8 (world = 8, the target of BASELINE configs[3]).
    sliced = 1: an all-to-all is registered as well, so the 4H- and K-sized transforms of rounds 2 and 3 and the pointwise
    work between them run on each rank's slices (mh_ntt_dist_dev inside the prover) -- same bytes; with 3 ranks
    (not a power of two) the prover stays on the replicated rounds."""
    import subprocess, sys
    a, b = 0x1234567, 0x7654321
    n = 1 << log_n
    srs = GM.universal_setup(n, n, 3 * n, TAU, GAMMA, pc=pc)
    nc, ni, mats, inst, wit = GM.dummy_circuit(a, b, 10, n)
    pk = GM.index(srs, nc, ni, mats, pc=pc)
    want = GM.prove(pk, inst, wit, bytes(range(32))) + GM.prove(pk, inst, wit, bytes(range(1, 33)))
    script = tmp_path / "shard_worker.py"
    script.write_text(SHARD_WORKER % {"root": ROOT, "out": str(tmp_path), "tau": TAU, "gamma": GAMMA, "a": a, "b": b, "log_n": log_n, "pc": pc,
                                      "sliced": sliced})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29613 + world + log_n + 40 * sliced), WORLD_SIZE=str(world), MH_CHECK="1")
    if sliced:
        env["MH_SLICED"] = "2"               # also with 2 ranks, where the library would keep the rounds replicated (not worth the bytes)
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r))) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    for r in range(world):
        assert open(tmp_path / ("proof%d.bin" % r), "rb").read() == want


SKEW_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np
import torch.distributed as dist
import marlin_amd as M
from marlin_amd import dist as MD
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
M.init(0)
n = %(n)d
one = M.api.FR_ONE_MONT
tau = np.array([0x1234567, 0, 0, 0], dtype=np.uint64)          # any field element serves as tau here
B = M.Bases.srs_powers(tau, n)
B.precompute(%(c)d)
rng = np.random.default_rng(7)
dense = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
hot = dense.copy()
hot[: n - n // 8] = np.array([0x1234, 0, 0, 0], dtype=np.uint64)   # Montgomery words: ONE non-zero digit, bucket 0x1233
d_dense, d_hot = M.DeviceBuffer.from_numpy(dense), M.DeviceBuffer.from_numpy(hot)
jobs = [(B, 0, d_hot, n), (B, 0, d_dense, n), (B, 5, d_hot, n - 5)]
single = M.msm_batch_dev(jobs, montgomery=True)
fb0, vb0 = M.msm_path_counts()
MD.enable_sharded_prove(dist)
sharded = M.msm_batch_sharded_dev(jobs, montgomery=True)
fb1, vb1 = M.msm_path_counts()
aff = lambda a: [tuple(M.g1_to_affine(r)[0]) for r in a]
assert aff(single) == aff(sharded), "rank %%d: sharded result differs" %% rank
open(os.path.join(%(out)r, "r%%d.txt" %% rank), "w").write("%%d %%d" %% (fb1 - fb0, vb1 - vb0))
dist.barrier(); dist.destroy_process_group()
'''


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_msm_with_skewed_digits(gpu, tmp_path, world):
    """ADVICE r02 (capi.hip skew fallback x sharding): 7/8 of the scalars are ONE value whose Montgomery words have a single
    non-zero digit, so 7/8 of the entries fall into one bucket -- which lives in exactly one rank's partitions.  Only that
    rank's largest-bucket test trips: it alone leaves the fixed-base path and computes the WHOLE sums of the group on the
    variable-base path, while its peers return shares.  The share / whole flags in the all_gather payload make every rank
    take the whole results.  Checked against the unsharded batch on every rank; the path counters show that the ranks really
    did disagree."""
    import subprocess, sys
    script = tmp_path / "skew_worker.py"
    script.write_text(SKEW_WORKER % {"root": ROOT, "out": str(tmp_path), "n": 1 << 16, "c": 16})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29713 + world), WORLD_SIZE=str(world), MH_CHECK="1")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r))) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    counts = [tuple(int(x) for x in open(tmp_path / ("r%d.txt" % r)).read().split()) for r in range(world)]
    assert any(vb > 0 for _, vb in counts), counts           # some rank fell back to the variable-base path ...
    assert any(vb == 0 for _, vb in counts), counts          # ... and some rank did not: the decision differed across ranks


def _random_r1cs(nc, ni_raw, seed):
    """random satisfiable R1CS with arbitrary coefficients: rows A_i, B_i are random sparse combinations of the
    variables, C_i = (c_i / z_j) * e_j for a random non-zero variable j.  Returns an oracle R1CS (unpadded)."""
    import random
    R = F.R_MOD
    rng = random.Random(seed)
    ni = 1
    while ni < ni_raw:
        ni *= 2
    nvars = nc                                   # square already: nc variables in total
    inst = [1] + [rng.randrange(R) for _ in range(ni_raw - 1)]
    inst_p = inst + [0] * (ni - ni_raw)
    wit = [rng.randrange(1, R) for _ in range(nvars - ni)]
    z = inst_p + wit
    A, B, C = [], [], []
    for _ in range(nc):
        ra = sorted(set(rng.randrange(nvars) for _ in range(rng.randrange(0, 4))))
        rb = sorted(set(rng.randrange(nvars) for _ in range(rng.randrange(1, 3))))
        rowa = [(rng.randrange(1, R) if rng.random() < 0.7 else 1, j) for j in ra]
        rowb = [(rng.randrange(1, R) if rng.random() < 0.7 else 1, j) for j in rb]
        va = sum(f * z[j] for f, j in rowa) % R
        vb = sum(f * z[j] for f, j in rowb) % R
        j = rng.randrange(ni, nvars)             # a witness column (non-zero value)
        rowc = [(va * vb % R * pow(z[j], -1, R) % R, j)] if va * vb % R else []
        A.append(rowa); B.append(rowb); C.append(rowc)
    return AHP.R1CS(inst, wit, A, B, C), ni


def _csr(rows, with_vals=True):
    from tests.util import fr_to_np
    import numpy as np
    rp = np.zeros(len(rows) + 1, dtype=np.uint64)
    cols, vals = [], []
    for i, r in enumerate(rows):
        for f, j in r:
            cols.append(j); vals.append(f)
        rp[i + 1] = len(cols)
    v = fr_to_np(vals) if vals else np.zeros((0, 4), dtype=np.uint64)
    return rp, np.array(cols, dtype=np.uint32), v


@pytest.mark.parametrize("nc,ni_raw,seed", [(40, 3, 1), (64, 8, 2), (100, 5, 3)])
def test_random_r1cs_with_coefficients_matches_oracle(gpu, nc, ni_raw, seed):
    """General matrices (non-unit coefficients, empty rows, repeated columns across A/B/C, |X| up to 8):
    byte-identical proof and index commitments vs the oracle."""
    from tests.util import fr_to_np
    cs_raw, ni = _random_r1cs(nc, ni_raw, seed)
    cs = AHP.pad_and_square(cs_raw)
    assert cs.num_constraints == nc and len(cs.instance) == ni
    nnz_bound = 3 * 4 * nc
    srs_o = MR.universal_setup(nc, nc, nnz_bound, TAU, GAMMA)
    pk_o = MR.marlin_index(srs_o, cs)
    pr = MR.prove(pk_o, cs, FS.ChaChaRng(SEED, 20))
    assert MR.verify(pk_o, cs_raw.instance[1:], pr)
    srs = GM.universal_setup(nc, nc, nnz_bound, TAU, GAMMA)
    pk = GM.index(srs, nc, ni, [_csr(cs.a), _csr(cs.b), _csr(cs.c)])
    assert pk.vk_bytes() == MR.vk_bytes(pk_o)
    proof = GM.prove(pk, fr_to_np(cs.instance), fr_to_np(cs.witness), SEED)
    assert proof == MR.proof_bytes(pr)


def test_all_zero_matrix_a_poly_zip_quirk(gpu):
    """src/ahp/prover.rs:625-637 builds a_poly by zipping the coefficient vectors of val_a, val_b, val_c: with an all-zero C
    matrix val_c is the zero polynomial (an empty vector), the zip is empty and a_poly = 0.  The product instead evaluates
    the full combination on the coset g K.  Both give the same h_2 -- any a of degree < |K| only enters the remainder of
    the division by v_K, which the reference discards (686-689) -- so the proofs are byte-identical (the oracle mirrors
    the zip) and verify."""
    from tests.util import fr_to_np
    import random
    R = F.R_MOD
    rng = random.Random(11)
    nc, ni = 32, 2
    inst = [1, rng.randrange(R)]
    wit = [0] + [rng.randrange(1, R) for _ in range(nc - ni - 1)]          # variable ni is zero
    A = [[(rng.randrange(1, R), rng.randrange(nc))] for _ in range(nc)]
    B = [[(rng.randrange(1, R), ni)] for _ in range(nc)]                   # B z = 0 in every row
    C = [[] for _ in range(nc)]                                            # all-zero matrix
    cs_raw = AHP.R1CS(inst, wit, A, B, C)
    cs = AHP.pad_and_square(cs_raw)
    srs_o = MR.universal_setup(nc, nc, 3 * nc, TAU, GAMMA)
    pk_o = MR.marlin_index(srs_o, cs)
    assert len(pk_o.index.polys["c_val"]) == 0 and len(pk_o.index.polys["a_val"]) > 0     # the zip truncates to nothing
    pr = MR.prove(pk_o, cs, FS.ChaChaRng(SEED, 20))
    assert MR.verify(pk_o, cs_raw.instance[1:], pr)
    srs = GM.universal_setup(nc, nc, 3 * nc, TAU, GAMMA)
    pk = GM.index(srs, nc, ni, [_csr(cs.a), _csr(cs.b), _csr(cs.c)])
    assert pk.vk_bytes() == MR.vk_bytes(pk_o)
    proof = GM.prove(pk, fr_to_np(cs.instance), fr_to_np(cs.witness), SEED)
    assert proof == MR.proof_bytes(pr)


def test_prove_dev_equals_prove(gpu):
    """mh_marlin_prove_dev (assignment already in device memory -- what bench.py times) makes the same proof bytes as
    mh_marlin_prove (host pointers)."""
    n = 1 << 10
    srs = GM.universal_setup(n, n, 3 * n, TAU, GAMMA)
    nc, ni, mats, inst, wit = GM.dummy_circuit(0x1234567, 0x7654321, 10, n)
    pk = GM.index(srs, nc, ni, mats)
    want = GM.prove(pk, inst, wit, SEED)
    d_inst, d_wit = gpu.DeviceBuffer.from_numpy(inst), gpu.DeviceBuffer.from_numpy(wit)
    assert GM.prove_dev(pk, d_inst, d_wit, SEED) == want
    assert GM.prove_dev(pk, d_inst, d_wit, SEED) == want          # the inputs are not modified


def test_index_and_prove_error_paths(gpu):
    import numpy as np
    srs = GM.universal_setup(64, 64, 192, TAU, GAMMA)
    nc, ni, mats, inst, wit = GM.dummy_circuit(3, 5, 10, 64)
    with pytest.raises(gpu.MarlinHipError, match="power of two"):
        GM.index(srs, nc, 3, mats)                                    # InvalidPublicInputLength
    bad = [(m[0], m[1].copy(), m[2]) for m in mats]
    bad[0][1][0] = 10 ** 6
    with pytest.raises(gpu.MarlinHipError, match="column index"):
        GM.index(srs, nc, ni, bad)
    small = GM.universal_setup(8, 8, 24, TAU, GAMMA)
    with pytest.raises(gpu.MarlinHipError, match="IndexTooLarge"):
        GM.index(small, nc, ni, mats)
    pk = GM.index(srs, nc, ni, mats)
    with pytest.raises(AssertionError):
        GM.prove(pk, inst[:1], wit, SEED)
    from marlin_amd import _lib
    import ctypes as C
    out = (C.c_uint8 * 16)(); n = C.c_size_t()
    rc = _lib.load().mh_marlin_prove(pk.handle, inst.ctypes.data, wit.ctypes.data, SEED, 20, out, 16, C.byref(n))
    assert rc != 0 and n.value == GM.proof_bytes_len("marlin")                      # buffer too small: reports the needed size
    rc = _lib.load().mh_marlin_prove(pk.handle, inst.ctypes.data, wit.ctypes.data, SEED, 7, out, 16, C.byref(n))
    assert rc != 0                                                    # unsupported ChaCha round count
    rc = _lib.load().mh_marlin_prove(12345, inst.ctypes.data, wit.ctypes.data, SEED, 20, out, 16, C.byref(n))
    assert rc != 0


def test_config0_2p10_plumbing(gpu):
    """BASELINE.json configs[0]: DummyCircuit, 2^10 constraints, MarlinKZG10 -- still within the Python oracle's
    reach for the verifier; |H| = 2^10, |K| = 2^12, 21 NTTs <= 2^13, MSMs <= 4095 points."""
    from tests.verify_adapter import oracle_verify
    rng = FS.test_rng()
    a, b = FS.fr_rand(rng), FS.fr_rand(rng)
    n = 1 << 10
    srs = GM.universal_setup(n, n, 3 * n, TAU, GAMMA)
    ncp, ni, mats, inst, wit = GM.dummy_circuit(a, b, 10, n)
    pk = GM.index(srs, ncp, ni, mats)
    assert (pk.H, pk.K, pk.X) == (1 << 10, 1 << 12, 2)
    proof = GM.prove(pk, inst, wit, SEED)
    assert oracle_verify(pk.vk_bytes(), srs.max_degree, TAU, GAMMA, [a * b % F.R_MOD], proof)


# ---- second PC path: SonicKZG10 (benches/bench.rs:81) -------------------------------------------------------
@pytest.mark.parametrize("kind,nc,nv", [("dummy", 32, 10), ("dummy", 128, 10), ("test", 25, 25), ("test", 100, 25)])
def test_sonic_proof_matches_oracle(gpu, kind, nc, nv):
    """SonicKZG10: index commitments and proof are byte-identical to the oracle's restatement, verify, and reject a
    wrong input."""
    rng = FS.test_rng()
    a, b = FS.fr_rand(rng), FS.fr_rand(rng)
    if kind == "dummy":
        cs = AHP.pad_and_square(AHP.dummy_circuit(a, b, nv, nc))
        ncp, ni, mats, inst, wit = GM.dummy_circuit(a, b, nv, nc)
        pub = [a * b % F.R_MOD]
    else:
        cs = AHP.pad_and_square(AHP.finalize_test_circuit(AHP.test_circuit(a, b, nc, nv)))
        ncp, ni, mats, inst, wit = GM.test_circuit(a, b, nc, nv)
        pub = [a * b % F.R_MOD, a * b % F.R_MOD * b % F.R_MOD]
    m = max(nc, nv)
    srs_o = MR.universal_setup(m, m, 3 * m, TAU, GAMMA)
    pk_o = MR.marlin_index(srs_o, cs, "sonic")
    pr = MR.prove(pk_o, cs, FS.ChaChaRng(SEED, 20))
    assert MR.verify(pk_o, pub, pr) and not MR.verify(pk_o, [a] * len(pub), pr)
    srs = GM.universal_setup(m, m, 3 * m, TAU, GAMMA, pc="sonic")
    pk = GM.index(srs, ncp, ni, mats, pc="sonic")
    assert pk.vk_bytes() == MR.vk_bytes(pk_o)
    proof = GM.prove(pk, inst, wit, SEED)
    assert len(proof) == GM.proof_bytes_len("sonic")
    assert proof == MR.proof_bytes(pr)


def test_sonic_full_size_verifies(gpu):
    """BASELINE.json configs[4] shape: 2^20 constraints, SonicKZG10 (on whichever curve this process runs)."""
    from tests.verify_adapter import oracle_verify
    rng = FS.test_rng()
    a, b = FS.fr_rand(rng), FS.fr_rand(rng)
    n = 1 << 20
    srs = GM.universal_setup(n, n, 3 * n, TAU, GAMMA, pc="sonic")
    ncp, ni, mats, inst, wit = GM.dummy_circuit(a, b, 10, n)
    pk = GM.index(srs, ncp, ni, mats, pc="sonic")
    proof = GM.prove(pk, inst, wit, SEED)
    vk = pk.vk_bytes()
    assert oracle_verify(vk, srs.max_degree, TAU, GAMMA, [a * b % F.R_MOD], proof, pc="sonic")
    assert not oracle_verify(vk, srs.max_degree, TAU, GAMMA, [a], proof, pc="sonic")


def test_bench_gpus_2_runs_two_sharded_ranks(gpu):
    """`python bench.py --gpus 2` launches two ranks that prove with sharded MSMs (both on this box's one GPU, gloo
    exchange) and reports n_gpus = 2 with a real number."""
    import json, subprocess, sys
    env = dict(os.environ, BENCH_BACKEND="gloo", BENCH_SINGLE_DEVICE="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearsal", "--steps", "1", "--warmup", "1",
                          "--log-constraints", "14", "--no-cpu-baseline", "--no-throughput"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["n_gpus"] == 2 and rec["value"] > 0 and rec["config"]["constraints"] == 1 << 14
    # the line carries what was produced: both ranks hold the same proof, the product's verifier accepts it, and it is the
    # proof one GPU makes from the same inputs and seed
    assert rec["proof"]["verified"] is True and rec["proof"]["identical_on_all_ranks"] is True, rec["proof"]
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--log-constraints", "14",
                          "--no-cpu-baseline", "--no-throughput", "--no-seam-route"], env=env, capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-3000:]
    rec1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    assert rec1["proof"]["verified"] is True and rec1["proof"]["sha256_32"] == rec["proof"]["sha256_32"]
    assert rec["rehearsal"] is True and rec["distinct_devices"] == 1
    # without --rehearsal two ranks on ONE device are refused a scaling value: the line still explains itself
    bare = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--log-constraints", "12",
                           "--no-cpu-baseline", "--no-throughput", "--no-verify"], env=env, capture_output=True, text=True, timeout=600)
    assert bare.returncode == 0, bare.stderr[-3000:]
    recb = json.loads([l for l in bare.stdout.splitlines() if l.startswith("{")][-1])
    assert recb["value"] is None and recb["rehearsal"] is False and "not an N-GPU measurement" in recb["note"]


@pytest.mark.skipif(F.CURVE != "bls12_381", reason="the golden proofs are BLS12-381 + MarlinKZG10")
def test_bench_line_says_its_proof_is_the_oracles_golden_proof(gpu):
    """bench.py proves the inputs of tests/golden/marlin_proofs_xl.json (same trapdoors, witness values and zk seed): at a size
    that file holds, the line it prints states that its last proof is byte-identical to the CPU oracle's and verifies."""
    import json, subprocess, sys
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--log-constraints", "16",
                          "--no-cpu-baseline", "--no-throughput", "--no-seam-route"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["proof"]["verified"] is True and rec["proof"]["oracle_golden"]["byte_identical"] is True, rec["proof"]


@pytest.mark.parametrize("pc", ["marlin", "sonic"])
def test_prove_with_caller_supplied_zk_draws(gpu, pc):
    """`Marlin::prove` is generic over `R: RngCore` (src/lib.rs:151-155).  mh_marlin_prove_draws takes the field elements
    the caller's rng produced, in consumption order; fed with the draws of ChaCha20Rng(SEED) --
    generated independently by tests/zkstream.py -- it must return the very proof mh_marlin_prove(SEED) returns; one draw
    too few, or a draw >= r, is MH_EINVAL."""
    import numpy as np
    import marlin_amd as M
    from tests import zkstream as ZS
    n = 1 << 12
    srs = GM.universal_setup(n, n, 3 * n, TAU, GAMMA, pc=pc)
    ncp, ni, mats, inst, wit = GM.dummy_circuit(0x1234567, 0x7654321, 10, n)
    pk = GM.index(srs, ncp, ni, mats, pc=pc)
    want = GM.prove(pk, inst, wit, SEED)
    nd = GM.zk_draw_count(pk)
    assert nd == 3 + 3 * pk.H + (12 if pc == "sonic" else 15)
    draws = ZS.fr_draws(SEED, nd)
    assert GM.prove_draws(pk, inst, wit, draws) == want
    with pytest.raises(M.MarlinHipError):
        GM.prove_draws(pk, inst, wit, draws[:-1])
    bad = draws.copy()
    bad[5] = np.array([0xffffffffffffffff] * 4, dtype=np.uint64)
    with pytest.raises(M.MarlinHipError):
        GM.prove_draws(pk, inst, wit, bad)


def test_bench_gpus_4_runs_the_sliced_rounds(gpu):
    """`python bench.py --gpus 4` (four ranks on this box's one GPU, gloo exchange): the all-to-all self-test passes on every
    rank, rounds 2 and 3 and the openings run on slices, and the line says so."""
    import json, subprocess, sys
    env = dict(os.environ, BENCH_BACKEND="gloo", BENCH_SINGLE_DEVICE="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--rehearsal", "--steps", "1", "--warmup", "1",
                          "--log-constraints", "14", "--no-cpu-baseline", "--no-throughput"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["n_gpus"] == 4 and rec["value"] > 0 and "slices" in rec["config"]["parallelism"], rec["config"]
    assert len(rec["ranks_seen"]) == 4 and rec["distinct_devices"] == 1          # four ranks, one physical GPU on this box
    assert rec["proof"]["verified"] is True and rec["proof"]["identical_on_all_ranks"] is True, rec["proof"]
    # a scaling record explains itself: every rank's own step time, kernel families and what the exchanges cost it
    assert rec["transport"]["kind"] == "callback-torch.distributed-gloo" and rec["transport"]["native_rccl"] is None
    assert sorted(r["rank"] for r in rec["per_rank"]) == [0, 1, 2, 3]
    for r in rec["per_rank"]:
        b = r["breakdown_ms_per_step"]
        assert r["ms_per_step"] > 0 and b["msm_accum"] > 0 and b["exchanges_per_step"] >= 11 and b["exchange_host_wall"] > 0, r


RCCL_WORKER = r'''
import os, sys, ctypes as C
sys.path.insert(0, %(root)r)
import numpy as np
import torch, torch.distributed as dist
import marlin_amd as M
from marlin_amd import dist as MD, _lib
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
M.init(0)
MD.enable_sharded_prove(dist, device=torch.device("cuda", 0))
send = np.arange(1, 41, dtype=np.uint64)                 # the size of a two-job payload: 2 x 18 limbs + error word + flags
recv = np.zeros(40, dtype=np.uint64)
for _ in range(3):                                       # persistent staging buffers are reused
    recv[:] = 0
    _lib.check(_lib.load().mh_marlin_probe_allgather(send.ctypes.data, send.nbytes, recv.ctypes.data), "test_allgather")
    assert np.array_equal(send, recv)
t = torch.ones(8, device="cuda"); dist.all_reduce(t); assert float(t.sum()) == 8.0
# the all-to-all of the distributed transforms: the registered callback on two of the library's own device buffers
# (zero-copy views of them go straight into RCCL), twice so that the cached views are reused
MD.enable_alltoall(dist, device=torch.device("cuda", 0))
x = np.random.default_rng(5).integers(0, 1 << 62, size=(1 << 14, 4), dtype=np.uint64)
a, b = M.DeviceBuffer.from_numpy(x), M.DeviceBuffer(x.nbytes)
for _ in range(2):
    assert MD._keepalive["a2a"](a.ptr, x.nbytes, b.ptr, None) == 0
    assert np.array_equal(b.download(x.shape), x)
    b.upload(np.zeros_like(x))
assert MD._keepalive["a2a_state"]["zero_copy"], "RCCL ran on staging tensors, not on the library's buffers"
# stream-ordered: the library on a torch stream, the collective enqueued under it, no host synchronisation inside the callback;
# the download that follows is ordered behind it on the same stream.  A proof made on that stream is the same proof.
from marlin_amd import marlin as GM
n = 1 << 12
srs = GM.universal_setup(n, n, 3 * n, 0x1234567, 0x7654321)
ncp, ni, mats, inst, wit = GM.dummy_circuit(3, 5, 10, n)
pk = GM.index(srs, ncp, ni, mats)
want = GM.prove(pk, inst, wit, bytes(range(32)))
ts = MD.use_torch_stream(torch.device("cuda", 0))
MD.enable_alltoall(dist, device=torch.device("cuda", 0), stream=ts)
for _ in range(3):
    b.upload(np.zeros_like(x))
    assert MD._keepalive["a2a"](a.ptr, x.nbytes, b.ptr, None) == 0
    assert np.array_equal(b.download(x.shape), x)
st = MD._keepalive["a2a_state"]
assert st["zero_copy"] and st["stream_ordered"] and st["calls"] == 3, st
assert GM.prove(pk, inst, wit, bytes(range(32))) == want
print("rccl world=1 ok:", torch.cuda.get_device_name(0))
dist.destroy_process_group()
'''


def test_exchange_callback_over_rccl_world_1(gpu, tmp_path):
    """The transport bench.py uses on a multi-GPU node -- torch.distributed backend "nccl" = RCCL, device tensors, the
    persistent pinned staging of marlin_amd/dist.py -- executed for real on this one-GPU box as a communicator of ONE rank
    (RCCL refuses two ranks on one device, so the N > 1 tests use gloo): the library's exchange hook runs the registered
    callback, the payload comes back intact.  A wrong stream, dtype or buffer handling in the RCCL branch fails here."""
    import subprocess, sys
    script = tmp_path / "rccl_worker.py"
    script.write_text(RCCL_WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29791", WORLD_SIZE="1", RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "rccl world=1 ok" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


def test_bench_line_carries_the_fine_print(gpu):
    """the default line itself says what the host-pointer entry point costs (`host_inputs_ms_per_step`), how
    the bucket reduction stands against its issue bound (`roofline_reduce`), whether the PMC capture is of the loaded build, and -- apart
    from the headline and only when asked for (`--throughput`) -- what two independent provers sharing the GPU deliver
    (`throughput_pipelined`)."""
    import json, subprocess, sys
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--log-constraints", "14",
                          "--no-cpu-baseline", "--no-seam-route", "--throughput"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["value"] > 0 and rec["host_inputs_ms_per_step"] > 0 and "HOST pointers" in rec["host_inputs_note"]
    rr = rec["roofline_reduce"]
    assert rr["bound"] == "valu-issue" and 0 < rr["frac"] < 1 and rr["additions_per_bucket"] == 2 and "rsum_kernel" in rr["kernel"]
    assert "capture_is_of_the_loaded_build" in rec["roofline"]["traffic_source"]
    tp = rec["throughput_pipelined"]
    assert tp["processes"] == 2 and tp["proofs_per_s"] > 0 and len(tp["proofs_per_process"]) == 2 and "not the headline" in tp["what"]
