"""CPU tests of the product's host-side logic (marlin_amd/csrc/host_ff.h, fs_host.h) through the
host-only hook library libmarlin_hosttest.so, against hashlib and the oracle."""
import ctypes as C
import hashlib
import os
import numpy as np
from oracle import fields as F, curve as EC, fs as FS, marlin as MR
from tests.util import fr_to_np, np_to_fr, fq_to_limbs, limbs_to_fq

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = C.CDLL(os.path.join(ROOT, "marlin_amd", "libmarlin_hosttest.so"))


def test_blake2s_matches_hashlib():
    for n in [0, 1, 3, 63, 64, 65, 127, 128, 129, 1000]:
        data = bytes((i * 7 + n) & 0xFF for i in range(n))
        out = (C.c_uint8 * 32)()
        L.ht_blake2s(data, C.c_size_t(n), out)
        assert bytes(out) == hashlib.blake2s(data).digest(), n


def test_chacha_and_fr_rand_match_oracle():
    seed = bytes(range(32))
    for rounds in (20, 12):
        out = np.zeros(200, dtype=np.uint64)
        L.ht_chacha_u64(seed, rounds, C.c_size_t(200), C.c_void_p(out.ctypes.data))
        r = FS.ChaChaRng(seed, rounds)
        assert [int(x) for x in out] == [r.next_u64() for _ in range(200)]
    out = np.zeros((50, 4), dtype=np.uint64)
    L.ht_fr_rand(seed, 20, C.c_size_t(50), C.c_void_p(out.ctypes.data))
    r = FS.ChaChaRng(seed, 20)
    assert np_to_fr(out) == [FS.fr_rand(r) for _ in range(50)]


def test_fiat_shamir_matches_oracle():
    init, a1 = b"MARLIN-2019" + bytes(range(100)), bytes(range(200, 256)) * 7
    out = np.zeros((5, 4), dtype=np.uint64)
    L.ht_fs(init, C.c_size_t(len(init)), a1, C.c_size_t(len(a1)), C.c_void_p(out.ctypes.data))
    fs = FS.SimpleHashFiatShamirRng(init)
    fs.absorb(a1)
    want = [fs.rand_fr() for _ in range(4)] + [fs.rand_u128_as_fr()]
    assert np_to_fr(out) == want


def test_host_field_and_group_ops():
    a, b = 0x1234567890abcdef1234567890abcdef1234567890abcdef, F.R_MOD - 5
    x, y = fr_to_np([a])[0], fr_to_np([b])[0]
    r = np.zeros(4, dtype=np.uint64)
    L.ht_fr_mul(C.c_void_p(x.ctypes.data), C.c_void_p(y.ctypes.data), C.c_void_p(r.ctypes.data))
    assert np_to_fr(r)[0] == a * b % F.R_MOD
    L.ht_fr_inv(C.c_void_p(x.ctypes.data), C.c_void_p(r.ctypes.data))
    assert np_to_fr(r)[0] == pow(a, -1, F.R_MOD)
    qa, qb = F.Q_MOD - 12345, 0xdeadbeef << 300
    xa = np.array(fq_to_limbs(qa), dtype=np.uint64); xb = np.array(fq_to_limbs(qb), dtype=np.uint64)
    rq = np.zeros(6, dtype=np.uint64)
    L.ht_fq_mul(C.c_void_p(xa.ctypes.data), C.c_void_p(xb.ctypes.data), C.c_void_p(rq.ctypes.data))
    assert limbs_to_fq(rq) == qa * qb % F.Q_MOD
    g = np.array(fq_to_limbs(EC.G1_GEN[0]) + fq_to_limbs(EC.G1_GEN[1]), dtype=np.uint64)
    for k in [1, 2, 0xabcdef123456789, F.R_MOD - 1]:
        kk = np.array(F.to_limbs64(k, 4), dtype=np.uint64)
        out = np.zeros(12, dtype=np.uint64); inf = C.c_int()
        L.ht_g1_mul(C.c_void_p(g.ctypes.data), C.c_void_p(kk.ctypes.data), C.c_void_p(out.ctypes.data), C.byref(inf))
        assert (limbs_to_fq(out[:6]), limbs_to_fq(out[6:])) == EC.scalar_mul(EC.G1_GEN, k)
    # commitment ToBytes layout == oracle's
    p2 = EC.scalar_mul(EC.G1_GEN, 77)
    xy2 = np.array(fq_to_limbs(p2[0]) + fq_to_limbs(p2[1]), dtype=np.uint64)
    out = (C.c_uint8 * 195)()
    L.ht_put_commitment(C.c_void_p(g.ctypes.data), 0, None, out)
    assert bytes(out) == MR.commitment_bytes((EC.G1_GEN, None))
    L.ht_put_commitment(C.c_void_p(g.ctypes.data), 1, C.c_void_p(xy2.ctypes.data), out)
    assert bytes(out) == MR.commitment_bytes((EC.G1_GEN, (p2,)))


def test_fq30_constants_are_current_and_consistent(tmp_path):
    """fq30_consts.inc (30-bit-limb base field of the fixed-base accumulation) is what gen_fq30.py generates, and
    its numbers satisfy the identities the device code relies on."""
    import re
    import subprocess
    import sys
    csrc = os.path.join(ROOT, "marlin_amd", "csrc")
    out = tmp_path / "fq30.inc"
    out2 = tmp_path / "fq30_mul.inc"
    subprocess.run([sys.executable, os.path.join(csrc, "gen_fq30.py"), str(out), str(out2)], check=True)
    txt = out.read_text()
    assert txt == open(os.path.join(csrc, "fq30_consts.inc")).read()
    assert out2.read_text() == open(os.path.join(csrc, "fq30_mul_gen.inc")).read()

    def arr(block, name):
        m = re.search(r"%s\[\d+\] = \{([^}]*)\}" % name, block)
        return [int(x.strip().rstrip("u"), 16) for x in m.group(1).split(",")]

    for curve, p, l32 in (("BLS12_381", 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab, 12),
                          ("BN254", 21888242871839275222246405745257275088696311157297823662689037894645226208583, 8)):
        block = txt[txt.index("struct Fq30Params_" + curve):]
        block = block[:block.index("\n};")]
        nl = int(re.search(r"NL = (\d+)", block).group(1))
        val30 = lambda v: sum(x << (30 * i) for i, x in enumerate(v))
        val32 = lambda v: sum(x << (32 * i) for i, x in enumerate(v))
        assert val30(arr(block, "P")) == p and all(x < 1 << 30 for x in arr(block, "P"))
        for k in (2, 3, 4, 8):
            assert val30(arr(block, "P%d" % k)) == k * p
        pinv = int(re.search(r"PINV = 0x([0-9a-f]+)u", block).group(1), 16)
        pinv_pos = int(re.search(r"PINV_POS = 0x([0-9a-f]+)u", block).group(1), 16)
        assert (pinv * p + 1) % (1 << 30) == 0 and (pinv_pos * p) % (1 << 30) == 1
        r30, r32 = 1 << (30 * nl), 1 << (32 * l32)
        assert val30(arr(block, "ONE")) == r30 % p
        # mont32(x * R32, TO30) = x * R30 ; mont32(x * R30, FROM30) = x * R32   (mont32(a, b) = a b / R32)
        x = 0x1234567890abcdef1234567890abcdef % p
        assert (x * r32 % p) * val32(arr(block, "TO30")) * pow(r32, -1, p) % p == x * r30 % p
        assert (x * r30 % p) * val32(arr(block, "FROM30")) * pow(r32, -1, p) % p == x * r32 % p
        # lazy-reduction head room: the product of two values below 20 p stays below 2 p
        assert (20 * p) * (20 * p) // r30 + p < 2 * p


def test_zkstream_numpy_matches_sequential_oracle():
    """tests/zkstream.py (vectorised ChaCha + rejection sampling, used by the full-size GPU parity tests) reproduces
    oracle/fs.py's sequential ChaChaRng / fr_rand stream, across a 64-word buffer boundary and for both round counts."""
    from tests import zkstream as ZS
    seed = bytes(range(32))
    for rounds in (20, 12):
        r = FS.ChaChaRng(seed, rounds)
        words = ZS.chacha_words(seed, rounds, 0, 12).reshape(-1)
        assert [int(w) for w in words] == [r.next_u32() for _ in range(12 * 16)]
        r = FS.ChaChaRng(seed, rounds)
        want = [FS.fr_rand(r) for _ in range(700)]
        assert [ZS.mont_to_canonical(row) for row in ZS.fr_draws(seed, 700, rounds)] == want
    # the draws of one prove at |H| = 32 in Appendix-C order, incl. the mask polynomial's sum-over-H fix (prover.rs:373-380)
    H = 32
    r = FS.ChaChaRng(seed, 20)
    d = ZS.prove_zk_draws(seed, H)
    assert [d["r_w"], d["r_za"], d["r_zb"]] == [FS.fr_rand(r) for _ in range(3)]
    mask = [FS.fr_rand(r) for _ in range(3 * H)]
    mask[0] = (mask[0] - mask[0] - mask[H] - mask[2 * H]) % F.R_MOD
    assert np_to_fr(d["mask"]) == mask
    for name in ("blind_w", "blind_za", "blind_zb", "blind_g1", "blind_g1_shifted"):
        assert d[name] == [FS.fr_rand(r) for _ in range(3)]


def test_workload_inventories_are_consistent():
    """marlin_amd/workload.py restates SURVEY Appendix A as data (bench.py, the CPU baseline and the seam route all run these
    lists): 30 transforms / 15 MSMs of the reference against the 16 / 13 the prover executes, and -- for the seam -- how many
    elements each transform's Vec holds BEFORE ark-poly zero-pads it (mh_ntt_len uploads only those)."""
    from marlin_amd import workload as W
    for lg in (10, 16, 20):
        H = 1 << lg
        K = 4 * H
        inv, ex, lens = W.ntt_inventory(H), W.ntt_executed(H), W.ntt_input_lengths(H)
        assert len(inv) == 30 and len(ex) == 16 and len(lens) == 30
        assert all(0 < l <= (1 << lg_n) for l, (lg_n, _, _) in zip(lens, inv))
        # inverse transforms arrive full; at least the nine `const * v_H` factors and the z_a z_b / q_1 / b f factors arrive short
        assert all(l == (1 << lg_n) for l, (lg_n, inverse, _) in zip(lens, inv) if inverse)
        assert sum(1 for l, (lg_n, inverse, _) in zip(lens, inv) if not inverse and l < (1 << lg_n)) >= 14
        up, full = sum(32 * l for l in lens), sum(32 << lg_n for lg_n, _, _ in inv)
        assert 0.55 < up / full < 0.65                       # 1.78 GB instead of 2.95 GB at 2^20
        big, small = W.msm_inventory(H)
        assert len(big) == 15 and len(W.msm_executed(H)) == 13 and len(W.msm_executed(H, pc="sonic")) == 11
        assert sum(n for n, _ in W.msm_executed(H)) < sum(n for n, _ in big)
        assert W.executed_ntt_bytes(H) < W.algorithmic_bytes(H)[0]


def test_native_transport_reports_a_missing_librccl_instead_of_crashing():
    """with no librccl to be found (MH_RCCL_LIB names a file that does not exist) mh_rccl_unique_id must return
    an error with the loader's message -- rccl_native.h used to call dlerror() twice and built a std::string from the NULL the
    second call returns (SIGSEGV on exactly the path meant to degrade gracefully) -- and marlin_amd.dist.enable_native_rccl must
    return False, which is what lets bench.py --transport auto fall back to the callback transport.  No GPU needed: the library
    is looked up before anything touches the device."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import numpy as np\n"
        "from marlin_amd import _lib, dist as MD\n"
        "lib = _lib.load()\n"
        "ident = np.zeros(128, dtype=np.uint8)\n"
        "rc = lib.mh_rccl_unique_id(ident.ctypes.data)\n"
        "msg = (lib.mh_last_error() or b'').decode()\n"
        "assert rc != 0 and 'not found' in msg and '/nonexistent/librccl.so' in msg, (rc, msg)\n"
        "assert MD.enable_native_rccl() is False\n"
        "print('clean', rc)\n" % root)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, MH_RCCL_LIB="/nonexistent/librccl.so"), capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0 and "clean" in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-1500:])
