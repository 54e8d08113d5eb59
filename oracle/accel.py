"""Optional C backend for the three operations that stop the pure-Python oracle at ~2^14 constraints (oracle only).

`enable()` swaps, inside the oracle package,
  * `poly.ntt`            (transforms of >= 2^10 points)   -> oracle/c/ref_hotpath.c `ref_ntt_mt`
  * `marlin.msm`          (>= 2^10 coefficients)           -> `ref_msm` (arkworks' Pippenger, one task per window)
  * `curve.srs_powers`    (KZG10::setup's powers)          -> `ref_srs_powers`
for their C restatements, which tests/test_oracle_c.py pins against the Python definitions they replace (naive DFT /
radix-2 NTT, naive MSM, known discrete logs) on both curves.  Everything else -- the AHP rounds, the polynomial
arithmetic between transforms, Fiat-Shamir, the PC layer's logic -- stays the Python restatement.  With it
`oracle.marlin.prove` reaches 2^16 .. 2^18 constraints in minutes, which is how tests/golden/marlin_proofs_xl.json is
made (tests/golden/make_golden.py xl): whole proofs from a CPU prover that shares no code with the device.

Nothing enables this implicitly: the default oracle is pure Python.
"""
import os
import numpy as np
from . import fields as F, poly as _poly, marlin as _marlin, curve as _curve, cref

_R = F.R_MOD
_on = False
_THREADS = 1
_MIN_LOG = 10


def _to_np(vals):
    """canonical ints -> (n,4) uint64 Montgomery"""
    buf = b"".join((v % _R).to_bytes(32, "little") for v in vals)
    a = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 4).copy()
    if len(a):
        cref.lib().ref_fr_to_mont(a.ctypes.data, len(a))
    return a


def _from_np(a):
    a = np.ascontiguousarray(a, dtype=np.uint64).copy()
    if len(a):
        cref.lib().ref_fr_from_mont(a.ctypes.data, len(a))
    b = a.tobytes()
    return [int.from_bytes(b[32 * i:32 * i + 32], "little") for i in range(len(a))]


_py_ntt = _poly.ntt
_py_msm = _marlin.msm
_py_srs_powers = _curve.srs_powers


def _ntt(vals, log_n, inverse=False):
    if log_n < _MIN_LOG:
        return _py_ntt(vals, log_n, inverse)
    n = 1 << log_n
    a = _to_np(list(vals) + [0] * (n - len(vals)))
    assert len(a) == n
    return _from_np(cref.ntt(a, inverse=inverse, threads=_THREADS))


class NpPowers:
    """powers_of_g as one numpy array (what the C MSM reads) that still answers `srs.powers_of_g[i]` and slices with the
    oracle's affine int tuples."""

    def __init__(self, arr):
        self.arr = arr

    def __len__(self):
        return len(self.arr)

    def _pt(self, row):
        L = F.FQ_LIMBS64
        x = sum(int(row[k]) << (64 * k) for k in range(L))
        y = sum(int(row[L + k]) << (64 * k) for k in range(L))
        return (F.fq_from_mont(x), F.fq_from_mont(y))

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self._pt(self.arr[k]) for k in range(*i.indices(len(self.arr)))]
        return self._pt(self.arr[i])

    def __iter__(self):
        return (self._pt(r) for r in self.arr)


def _srs_powers(tau, n, base=_curve.G1_GEN):
    if base != _curve.G1_GEN or n < 64:
        return _py_srs_powers(tau, n, base)
    return NpPowers(cref.srs_powers(tau, n, threads=_THREADS))


def _msm(srs, offset, coeffs):
    coeffs = list(coeffs)
    if len(coeffs) < (1 << _MIN_LOG) or not isinstance(srs.powers_of_g, NpPowers):
        return _py_msm(srs, offset, coeffs)
    lz = 0
    while lz < len(coeffs) and coeffs[lz] % _R == 0:
        lz += 1
    if lz == len(coeffs):
        return None
    xyz = cref.msm(srs.powers_of_g.arr[offset + lz: offset + len(coeffs)], _to_np(coeffs[lz:]), montgomery=True, threads=_THREADS)
    xy, inf = cref.g1_to_affine(xyz)
    if inf:
        return None
    L = F.FQ_LIMBS64
    return (F.fq_from_mont(sum(int(xy[k]) << (64 * k) for k in range(L))), F.fq_from_mont(sum(int(xy[L + k]) << (64 * k) for k in range(L))))


def enable(threads=None):
    global _on, _THREADS
    _THREADS = threads or (len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
    _poly.ntt, _marlin.msm, _curve.srs_powers = _ntt, _msm, _srs_powers
    _on = True


def disable():
    global _on
    _poly.ntt, _marlin.msm, _curve.srs_powers = _py_ntt, _py_msm, _py_srs_powers
    _on = False
