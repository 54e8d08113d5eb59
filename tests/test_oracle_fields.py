"""Pins the oracle's constants (public known answers) and cross-checks the
constants compiled into the device / host headers against them."""
import os
import re
from oracle import fields as F, curve as EC

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _is_probable_prime(n):
    if n < 2:
        return False
    for p in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
        if n % p == 0:
            return n == p
    d, s = n - 1, 0
    while d % 2 == 0:
        d //= 2
        s += 1
    for a in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
        x = pow(a, d, n)
        if x in (1, n - 1):
            continue
        for _ in range(s - 1):
            x = x * x % n
            if x == n - 1:
                break
        else:
            return False
    return True


def test_public_known_answers():
    assert F.R_MOD.bit_length() == 255 and _is_probable_prime(F.R_MOD)
    assert F.Q_MOD.bit_length() == 381 and _is_probable_prime(F.Q_MOD)
    assert (F.R_MOD - 1) % (1 << 32) == 0 and (F.R_MOD - 1) % (1 << 33) != 0
    assert pow(F.FR_TWO_ADIC_ROOT, 1 << 32, F.R_MOD) == 1 and pow(F.FR_TWO_ADIC_ROOT, 1 << 31, F.R_MOD) != 1
    assert F.FR_TWO_ADIC_ROOT == 0x16a2a19edfe81f20d09b681922c813b4b63683508c2280b93829971f439f0d2b
    assert F.FR_MONT_R == 0x1824b159acc5056f998c4fefecbc4ff55884b7fa0003480200000001fffffffe
    assert F.FR_INV64 == 0xfffffffeffffffff and F.FQ_INV64 == 0x89f3fffcfffcfffd
    assert F.FQ_MONT_R == 0x15f65ec3fa80e4935c071a97a256ec6d77ce5853705257455f48985753c758baebf4000bc40c0002760900000002fffd
    assert EC.is_on_curve(EC.G1_GEN)
    assert EC.scalar_mul(EC.G1_GEN, F.R_MOD - 1) == EC.neg(EC.G1_GEN)      # [r]G = O
    assert EC.add(EC.scalar_mul(EC.G1_GEN, F.R_MOD - 1), EC.G1_GEN) is None


def _arr(src, struct, name):
    m = re.search(r"struct %s \{(.*?)\n\};" % struct, src, re.S)
    body = m.group(1)
    a = re.search(r"%s\[\d+\] = \{(.*?)\}" % name, body, re.S).group(1)
    return [int(re.sub(r"[uUlL]+$", "", x), 16) for x in re.findall(r"0x[0-9a-fA-F]+[uUlL]*", a)]


def _join(vals, bits):
    return sum(v << (bits * i) for i, v in enumerate(vals))


def test_device_header_constants():
    src = open(os.path.join(ROOT, "marlin_amd", "csrc", "ff.cuh")).read()
    assert _join(_arr(src, "FrParams", "MOD"), 32) == F.R_MOD
    assert _join(_arr(src, "FrParams", "ONE"), 32) == F.FR_MONT_R
    assert _join(_arr(src, "FrParams", "R2"), 32) == F.FR_MONT_R2
    assert _join(_arr(src, "FqParams", "MOD"), 32) == F.Q_MOD
    assert _join(_arr(src, "FqParams", "ONE"), 32) == F.FQ_MONT_R
    assert _join(_arr(src, "FqParams", "R2"), 32) == F.FQ_MONT_R2
    assert "INV = 0x%08xu" % F.FR_INV32 in src and "INV = 0x%08xu" % F.FQ_INV32 in src


def test_host_header_constants():
    src = open(os.path.join(ROOT, "marlin_amd", "csrc", "host_ff.h")).read()
    assert _join(_arr(src, "FrP", "MOD"), 64) == F.R_MOD
    assert _join(_arr(src, "FrP", "ONE"), 64) == F.FR_MONT_R
    assert _join(_arr(src, "FrP", "R2"), 64) == F.FR_MONT_R2
    assert _join(_arr(src, "FqP", "MOD"), 64) == F.Q_MOD
    assert _join(_arr(src, "FqP", "ONE"), 64) == F.FQ_MONT_R
    assert _join(_arr(src, "FqP", "R2"), 64) == F.FQ_MONT_R2
