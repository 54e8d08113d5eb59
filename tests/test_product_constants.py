"""Constants the product carries on its own (no oracle import in the product path) agree with the oracle's."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHECK = r'''
import sys
sys.path.insert(0, %(root)r)
import numpy as np
from marlin_amd import marlin as GM
from oracle import g2 as G2
from tests.util import fq_to_limbs
h = G2.G2_GEN
assert G2.g2_is_on_curve(h)
want = np.array(sum([fq_to_limbs(c) for c in (h[0][0], h[0][1], h[1][0], h[1][1])], []), dtype=np.uint64)
assert np.array_equal(GM.g2_generator_mont(), want)
print("ok")
'''


def test_g2_generator_of_the_bench_verifier_key_is_the_standard_one():
    """bench.py verifies its last proof with the product's host verifier and needs an h in G2 for the verifier key:
    marlin_amd.marlin.g2_generator_mont() is the standard generator, on both curves (the curve is a per-process choice)."""
    for curve in ("bls12_381", "bn254"):
        env = dict(os.environ, MARLIN_AMD_CURVE=curve, ORACLE_CURVE=curve)
        r = subprocess.run([sys.executable, "-c", CHECK % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0 and "ok" in r.stdout, curve + ": " + r.stderr[-1500:]
