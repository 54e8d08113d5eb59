"""Adapter: verify a flat proof from the HIP prover with the oracle's verifier (oracle/marlin.py
verify: Fiat-Shamir replay + known-tau KZG identity).  O(1) group operations, so it works at
2^20 constraints where the Python oracle could never produce the proof itself."""
from oracle import fields as F, curve as EC, marlin as MR
from oracle.poly import Domain


class _LazyPowers:
    """srs.powers_of_g[i] = [tau^i]G computed on demand."""

    def __init__(self, tau):
        self.tau = tau

    def __getitem__(self, i):
        return EC.scalar_mul(EC.G1_GEN, pow(self.tau, i, F.R_MOD))


FQB = F.FQ_BYTES
G1B = 2 * FQB + 1


def _g1(b):
    x = int.from_bytes(b[0:FQB], "little")
    y = int.from_bytes(b[FQB:2 * FQB], "little")
    return None if b[2 * FQB] else (x, y)


def _comm_len(pc):
    return G1B if pc == "sonic" else 2 * G1B + 1


def _commitment(b, pc):
    if pc == "sonic":
        return (_g1(b[0:G1B]), "sonic")
    comm = _g1(b[0:G1B])
    has = b[G1B]
    sh = _g1(b[G1B + 1:2 * G1B + 1])
    return (comm, (sh,) if has else None)


def parse_vk(vk, pc="marlin"):
    nv = int.from_bytes(vk[0:8], "little"); nc = int.from_bytes(vk[8:16], "little"); nnz = int.from_bytes(vk[16:24], "little")
    cl = _comm_len(pc)
    comms = [_commitment(vk[24 + cl * i: 24 + cl * (i + 1)], pc) for i in range(6)]
    return nv, nc, nnz, comms


def parse_proof(pb, pc="marlin"):
    cl = _comm_len(pc)
    assert len(pb) == 9 * cl + 128 + 2 * (G1B + 33)
    pr = MR.Proof()
    cs = [_commitment(pb[cl * i: cl * (i + 1)], pc) for i in range(9)]
    pr.commitments = [cs[0:4], cs[4:7], cs[7:9]]
    o = cl * 9
    pr.evaluations = [int.from_bytes(pb[o + 32 * i: o + 32 * (i + 1)], "little") for i in range(4)]
    o += 128
    pr.pc_proof = []
    for _ in range(2):
        w = _g1(pb[o:o + G1B]); has = pb[o + G1B]; rv = int.from_bytes(pb[o + G1B + 1:o + G1B + 33], "little")
        pr.pc_proof.append((w, rv if has else None))
        o += G1B + 33
    return pr


def oracle_verify(vk_bytes, srs_max_degree, tau, gamma, public_input, proof_bytes, pc="marlin", use_pairing=False, wire=False):
    """wire=True: proof_bytes are the CanonicalSerialize bytes (mh_marlin_proof_serialize); use_pairing=True: the KZG
    equation is decided with the BLS12-381 pairing (oracle/pairing.py) instead of the known-tau identity."""
    nv, nc, nnz, comms = parse_vk(vk_bytes, pc)
    pk = MR.IndexKeys()
    pk.pc = pc
    idx = type("Idx", (), {})()
    idx.num_variables, idx.num_constraints, idx.num_non_zero = nv, nc, nnz
    idx.domain_h, idx.domain_k = Domain(nc), Domain(nnz)
    pk.index = idx
    pk.index_comms = comms
    srs = type("Srs", (), {})()
    srs.max_degree, srs.tau, srs.gamma = srs_max_degree, tau % F.R_MOD, gamma % F.R_MOD
    srs.g = EC.G1_GEN
    srs.gamma_g = EC.scalar_mul(EC.G1_GEN, gamma)
    srs.powers_of_g = _LazyPowers(tau)
    pk.srs = srs
    pr = MR.proof_deserialize(proof_bytes, pc) if wire else parse_proof(proof_bytes, pc)
    return MR.verify(pk, list(public_input), pr, use_pairing=use_pairing)
