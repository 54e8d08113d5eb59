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
# name: (modulus, limbs, largest lazy operand in units of the modulus)
FIELDS = {"BLS12_381_FR": (0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001, 9, 450),
          "BN254_FR": (21888242871839275222246405745257275088548364400416034343698204186575808495617, 9, 450),
          # the base fields: the table coordinates of the fixed-base MSM (MH_FB_SHOUP=1); operands ZZ, ZZZ <= 1.1 p, tested to 20 p
          "BLS12_381": (0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab, 13, 20),
          "BN254": (21888242871839275222246405745257275088696311157297823662689037894645226208583, 9, 20)}


def limbs(x, nl):
    return [(x >> (30 * i)) & M for i in range(nl)]


def val(l):
    return sum(v << (30 * i) for i, v in enumerate(l))


def shoup_product(a, w, wq, rbar):
    """(t, q) exactly as f30_mulshoup_* computes them; asserts the accumulator bounds the generated code depends on."""
    NL = len(rbar)
    al, wl, wql = limbs(a, NL), limbs(w, NL), limbs(wq, NL)
    acc = sum(al[i] * wql[NL - 1 - i] for i in range(NL))
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
        last = k == NL - 1                                  # the last column may wrap: only its low 30 bits are used
        acc += sum(al[i] * wl[k - i] for i in range(k + 1))
        assert last or acc < 1 << 64, k
        early = 0
        qr = sum(q[i] * rbar[k - i] for i in range(k + 1))
        if not last and acc + qr >= 1 << 64:                # the generator moves the carry out here whenever the BOUND says so
            early, acc = acc >> 30, acc & M
        acc = (acc + qr) & ((1 << 64) - 1) if last else acc + qr
        assert last or acc < 1 << 64, k
        t.append(acc & M)
        acc = (acc >> 30) + early
    return val(t), val(q)


@pytest.mark.parametrize("name", sorted(FIELDS))
def test_shoup_product_model(name):
    r, NL, lazy = FIELDS[name]
    BETA = 1 << (30 * NL)
    rbar, rinv = limbs(BETA - r, NL), pow(r, -1, BETA)
    rng = random.Random(20260926)
    worst = 0
    for it in range(4000):
        w = rng.randrange(1, r) if it else r - 1
        rho = w * BETA % r                                  # what tw30 holds (w R' mod r)
        wq = (BETA - rho) * rinv % BETA                     # build_twiddles30s
        assert rho != 0 and wq == w * BETA // r
        # lazy operands (NTT: up to (1 + 16 * 28) r = 449 r after 28 stages); low limbs normalised, all-ones limbs as the extreme
        a = rng.randrange(0, lazy * r) if it % 3 else lazy * r - 1 - rng.randrange(1000)
        if it % 7 == 0:
            a = val([M] * (NL - 1) + [(lazy * r) >> (30 * (NL - 1))])
        t, q = shoup_product(a, w, wq, rbar)
        big_q = a * wq >> (30 * NL)
        assert 0 <= big_q - q <= NL
        assert t == a * w - q * r and t % r == a * w % r and 0 <= t < (2 + NL) * r
        worst = max(worst, t // r)
    assert worst <= 1 + NL
    # NTT: a - t + 16 r never goes negative, and 28 stages of growth stay far below R' / r; MSM: 16 p - S2 likewise
    assert 16 * r >= (2 + NL) * r and (not name.endswith("_FR") or (1 + 16 * 28) * r < BETA >> 6)


def test_generated_constants_and_product_count():
    consts = open(os.path.join(ROOT, "marlin_amd", "csrc", "fq30_consts.inc")).read()
    gen = open(os.path.join(ROOT, "marlin_amd", "csrc", "fq30_mul_gen.inc")).read()
    for name, (r, NL, _) in FIELDS.items():
        BETA = 1 << (30 * NL)
        blk = consts.split("struct Fq30Params_%s {" % name)[1].split("\n};\n")[0]
        def arr(key):
            m = re.search(r"\b%s\[%d\] = \{([^}]*)\}" % (key, NL), blk)
            return [int(x.strip().rstrip("u"), 16) for x in m.group(1).split(",")]
        assert arr("RBAR") == limbs(BETA - r, NL)
        assert arr("PINV_FULL") == limbs(pow(r, -1, BETA), NL)
        assert arr("P16") == limbs(16 * r, NL) and arr("RR") == limbs(BETA * BETA % r, NL)
        fn = gen.split("void f30_mulshoup_%s(" % name)[1].split("\n}\n")[0]
        assert fn.count("v_mad_u64_u32") == 3 * NL * (NL + 1) // 2      # 135 against the Montgomery product's 162; 273 against 338
