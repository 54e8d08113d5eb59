"""Re-entrancy of the C ABI (the seams "must tolerate concurrent calls from rayon threads" -- the reference's
`cfg_iter!` / `cfg_into_iter!` sites, /root/reference src/ahp/prover.rs:340,473-477,537-544, run under rayon with the `parallel`
feature, so a Rust host that swaps `GeneralEvaluationDomain::fft` and `VariableBaseMSM::multi_scalar_mul` for the library's entry
points will call them from several threads at once).  The library serialises on its context (Context::mu); what is tested is that
concurrent callers get exactly the serial results -- no shared scratch is handed to two calls, no result lands in the wrong caller."""
import ctypes as C
import threading

import numpy as np
import pytest

from tests.util import fr_to_np

pytestmark = pytest.mark.gpu


def _rand_fr_np(rng, n):
    x = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
    x[:, 3] &= np.uint64((1 << 61) - 1)
    return x


def test_four_threads_on_one_context_get_the_serial_results(gpu):
    """Four Python threads (ctypes drops the GIL inside a call) issue mh_ntt_len, mh_msm and mh_msm_batch on DISTINCT inputs against
    the one context, 6 rounds each, interleaved however the scheduler likes: every result equals the one the same call returned
    alone beforehand."""
    from marlin_amd import _lib
    lib = _lib.load()
    n = 1 << 14
    B = gpu.Bases.srs_powers(fr_to_np([0x1234567])[0], 3 * n)
    B.precompute()
    XYZ = 3 * _lib.FQ_LIMBS

    def ntt_len(x, in_len, log_n, inverse):
        buf = np.zeros((1 << log_n, 4), dtype=np.uint64)
        buf[:in_len] = x[:in_len]
        _lib.check(lib.mh_ntt_len(_lib.CURVE_ID, buf.ctypes.data, in_len, log_n, inverse), "mh_ntt_len")
        return buf

    def msm(sc, off):
        out = np.zeros(XYZ, dtype=np.uint64)
        _lib.check(lib.mh_msm(B.handle, off, sc.ctypes.data, 1, sc.shape[0], out.ctypes.data), "mh_msm")
        return out

    def msm_batch(scs, offs):
        k = len(scs)
        handles = (C.c_uint64 * k)(*[B.handle] * k)
        o = (C.c_size_t * k)(*offs)
        ptrs = (C.c_void_p * k)(*[s.ctypes.data for s in scs])
        ns = (C.c_size_t * k)(*[s.shape[0] for s in scs])
        out = np.zeros((k, XYZ), dtype=np.uint64)
        _lib.check(lib.mh_msm_batch(k, handles, o, ptrs, ns, 1, out.ctypes.data), "mh_msm_batch")
        return out

    jobs = []
    for t in range(4):
        rng = np.random.default_rng(50 + t)
        x = _rand_fr_np(rng, 1 << 15)
        s1, s2, s3 = _rand_fr_np(rng, n + 17 * t), _rand_fr_np(rng, n), _rand_fr_np(rng, n // 2 + t)
        calls = [lambda x=x, t=t: ntt_len(x, (1 << 13) + 5 * t, 15, t & 1),
                 lambda s1=s1, t=t: msm(s1, 100 * t),
                 lambda s2=s2, s3=s3, t=t: msm_batch([s2, s3, s2], [t, n, 7 + t])]
        jobs.append(calls)
    # MSM results are compared as affine points: the Jacobian coordinates of one and the same sum depend on the order in which the
    # sort's LDS atomics happened to rank the entries of a bucket
    def canon(k, r):
        if k == 0:
            return np.array(r)
        return np.stack([gpu.g1_to_affine(row)[0] for row in np.atleast_2d(r)])
    want = [[canon(k, f()) for k, f in enumerate(calls)] for calls in jobs]          # serial reference
    errors = []

    def worker(t):
        try:
            for rnd in range(6):
                for k, f in enumerate(jobs[t]):
                    got = canon(k, f())
                    if not np.array_equal(got, want[t][k]):
                        errors.append((t, rnd, k))
        except Exception as e:                                            # noqa: BLE001
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=120)
    assert not any(th.is_alive() for th in threads), "a thread is stuck inside the library"
    assert errors == []


def test_msm_and_ntt_from_other_threads_while_a_proof_is_being_made(gpu):
    """One thread inside mh_marlin_prove (2^14 constraints, several proofs back to back) while two others call mh_msm and mh_ntt on
    their own data: the proofs are the ones made alone (the prover's scratch vectors, MSM workspace and side stream are not
    disturbed by calls that queue up behind it), and so are the other threads' results."""
    from marlin_amd import marlin as GM, _lib
    import json, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gold = json.load(open(os.path.join(root, "tests", "golden", "marlin_proofs.json")))
    tau, gamma = int(gold["tau"], 16), int(gold["gamma"], 16)
    n = 1 << 14
    srs = GM.universal_setup(n, n, 3 * n, tau, gamma)
    nc, ni, mats, inst, wit = GM.dummy_circuit(0x1234567, 0x7654321, 10, n)
    pk = GM.index(srs, nc, ni, mats)
    seeds = [bytes(range(k, k + 32)) for k in range(3)]
    want_proofs = [GM.prove(pk, inst, wit, s) for s in seeds]
    rng = np.random.default_rng(77)
    B = gpu.Bases.srs_powers(fr_to_np([0x7654321])[0], 1 << 13)        # another base set, with its own window table
    B.precompute()
    sc = _rand_fr_np(rng, 1 << 13)
    x = _rand_fr_np(rng, 1 << 16)
    want_msm, want_ntt = gpu.g1_to_affine(gpu.msm(B, sc))[0], gpu.ntt(x)
    errors, stop = [], threading.Event()

    def prover():
        try:
            for rnd in range(2):
                for s, w in zip(seeds, want_proofs):
                    if GM.prove(pk, inst, wit, s) != w:
                        errors.append(("proof", rnd))
        except Exception as e:                                            # noqa: BLE001
            errors.append(("prover", repr(e)))
        finally:
            stop.set()

    def other(which):
        try:
            count = 0
            while not stop.is_set() or count < 3:
                if which == 0:
                    ok = np.array_equal(gpu.g1_to_affine(gpu.msm(B, sc))[0], want_msm)
                else:
                    ok = np.array_equal(gpu.ntt(x), want_ntt)
                count += 1
                if not ok:
                    errors.append((("msm", "ntt")[which], count))
                    return
        except Exception as e:                                            # noqa: BLE001
            errors.append((which, repr(e)))

    threads = [threading.Thread(target=prover), threading.Thread(target=other, args=(0,)), threading.Thread(target=other, args=(1,))]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=180)
    assert not any(th.is_alive() for th in threads), "a thread is stuck inside the library"
    assert errors == []
