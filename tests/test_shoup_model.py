"""Integer model of the Shoup twiddle product of the NTT butterflies (marlin_amd/csrc/gen_fq30.py: shoup, ntt30.cuh:
butterfly_shoup, build_twiddles30s), limb for limb as the generated code computes it: the quotient from the columns i + j >= 8 of
b w', the result as the low nine limbs of b w + q (R' - r), the table entry as (R' - rho) r^-1 mod R'.  Checks what the kernel relies
on: no 64-bit accumulator overflows where its value is used, 0 <= t < 11 r, t = b w (mod r), and the constants the generator wrote."""
import os
import random
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
M = (1 << 30) - 1
NL = 9
BETA = 1 << (30 * NL)
FIELDS = {"BLS12_381_FR": 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001,
          "BN254_FR": 21888242871839275222246405745257275088548364400416034343698204186575808495617}


def limbs(x):
    return [(x >> (30 * i)) & M for i in range(NL)]


def val(l):
    return sum(v << (30 * i) for i, v in enumerate(l))


def shoup_product(a, w, wq, rbar):
    """(t, q) exactly as f30_mulshoup_* computes them; asserts the accumulator bounds the generated code depends on."""
    al, wl, wql = limbs(a), limbs(w), limbs(wq)
    acc = sum(al[i] * wql[8 - i] for i in range(NL))
    assert acc < 1 << 64
    acc >>= 30
    q = []
    for k in range(NL, 2 * NL - 1):
        acc += sum(al[i] * wql[k - i] for i in range(k - NL + 1, NL))
        assert acc < 1 << 64
        q.append(acc & M)
        acc >>= 30
    assert acc <= M
    q.append(acc)
    acc, t = 0, []
    for k in range(NL):
        acc += sum(al[i] * wl[k - i] for i in range(k + 1)) + sum(q[i] * rbar[k - i] for i in range(k + 1))
        assert k == NL - 1 or acc < 1 << 64, k          # the last column may wrap: only its low 30 bits are used
        acc &= (1 << 64) - 1
        t.append(acc & M)
        acc >>= 30
    return val(t), val(q)


@pytest.mark.parametrize("name", sorted(FIELDS))
def test_shoup_product_model(name):
    r = FIELDS[name]
    rbar, rinv = limbs(BETA - r), pow(r, -1, BETA)
    rng = random.Random(20260926)
    worst = 0
    for it in range(4000):
        w = rng.randrange(1, r) if it else r - 1
        rho = w * BETA % r                                  # what tw30 holds (w R' mod r)
        wq = (BETA - rho) * rinv % BETA                     # build_twiddles30s
        assert rho != 0 and wq == w * BETA // r
        # lazy operands: up to (1 + 16 * 28) r = 449 r after 28 stages; limbs 0..7 normalised, all-ones limbs as the extreme
        a = rng.randrange(0, 450 * r) if it % 3 else 450 * r - 1 - rng.randrange(1000)
        if it % 7 == 0:
            a = val([M] * 8 + [(450 * r) >> 240])
        t, q = shoup_product(a, w, wq, rbar)
        big_q = a * wq >> (30 * NL)
        assert 0 <= big_q - q <= NL
        assert t == a * w - q * r and t % r == a * w % r and 0 <= t < 11 * r
        worst = max(worst, t // r)
    assert worst <= 10
    # a - t + 16 r never goes negative, and 28 stages of growth stay far below R' / r
    assert 16 * r > 11 * r and (1 + 16 * 28) * r < BETA >> 6


def test_generated_constants_and_product_count():
    consts = open(os.path.join(ROOT, "marlin_amd", "csrc", "fq30_consts.inc")).read()
    gen = open(os.path.join(ROOT, "marlin_amd", "csrc", "fq30_mul_gen.inc")).read()
    for name, r in FIELDS.items():
        blk = consts.split("struct Fq30Params_%s {" % name)[1].split("\n};\n")[0]
        def arr(key):
            m = re.search(r"%s\[9\] = \{([^}]*)\}" % key, blk)
            return [int(x.strip().rstrip("u"), 16) for x in m.group(1).split(",")]
        assert arr("RBAR") == limbs(BETA - r)
        assert arr("PINV_FULL") == limbs(pow(r, -1, BETA))
        assert arr("P16") == limbs(16 * r)
        fn = gen.split("void f30_mulshoup_%s(" % name)[1].split("\n}\n")[0]
        assert fn.count("v_mad_u64_u32") == 135              # 45 + 45 + 45 against the Montgomery product's 162
