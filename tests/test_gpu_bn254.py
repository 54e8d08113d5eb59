"""Second curve (BASELINE.json configs[4]): the same sources built with -DMH_CURVE_BN254
(libmarlin_hip_bn254.so) pass the same kernel-level parity tests against the oracle restated over BN254
(y^2 = x^3 + 3, generator (1, 2), 254-bit r and q, two-adicity 28).  The curve is selected per process
(MARLIN_AMD_CURVE / ORACLE_CURVE), so the suite runs in a subprocess."""
import os
import subprocess
import sys
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(files, extra=()):
    env = dict(os.environ, MARLIN_AMD_CURVE="bn254", ORACLE_CURVE="bn254")
    cmd = [sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"] + list(extra) + files
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    return r.stdout


def test_bn254_ntt_msm_srs_parity():
    out = _run(["tests/test_gpu_ntt.py", "tests/test_gpu_msm.py", "tests/test_gpu_srs.py", "tests/test_gpu_g2.py"])
    assert " passed" in out


def test_bn254_marlin_and_sonic_provers():
    """The whole prover on BN254 with both PC schemes: byte-identical proofs vs the oracle (fresh runs), polynomial
    parity, general R1CS, proofs up to 2^16 verified by the oracle (the 2^20 proofs of BASELINE.json configs[4] are byte-pinned by
    test_bn254_whole_golden_proofs_of_the_cpu_oracle and recomputed by test_bn254_sonic_2p20_whole_proof_pinned)."""
    # (the multi-rank cases are transport and sharding logic, which is curve-independent and runs in full on BLS12-381: one of
    #  them -- 4 ranks on slices, buckets cut into parts -- is kept here; the bench.py launches, the switches' cross-check paths, the
    #  caller's Fiat-Shamir / zk draws, the error paths and the skewed sharded MSMs -- host logic and curve-generic kernels -- are not repeated)
    out = _run(["tests/test_gpu_marlin.py"],
               extra=["-k", "not golden and not two_ranks and not bench_gpus and not bench_line and not (full_size and 20) and not (sharded_prove_ranks and not "
                            "4-16-marlin-1) and not alternative_paths and not skewed_digits and not exchange_callback and not callers_fiat_shamir "
                            "and not zk_draws and not error_paths"])
    assert " passed" in out


def test_bn254_sonic_2p20_whole_proof_pinned():
    """BASELINE configs[4] byte-pinned: every commitment, evaluation and opening of a 2^20-constraint BN254 + SonicKZG10
    proof recomputed on the CPU (tests/cpu_open.py over libref_hotpath_bn254.so)."""
    out = _run(["tests/test_gpu_parity_pins.py"], extra=["-k", "bn254_sonic"])
    assert "1 passed" in out


def test_bn254_whole_golden_proofs_of_the_cpu_oracle():
    """BN254 + SonicKZG10 at 2^16 and 2^20 (BASELINE configs[4]) and BN254 + MarlinKZG10 at 2^20: the device's proofs are
    byte for byte the CPU oracle's golden proofs (tests/golden/marlin_proofs_xl_bn254_*.json, make_golden.py xl-cfg)."""
    import glob
    if not glob.glob(os.path.join(ROOT, "tests", "golden", "marlin_proofs_xl_bn254_*.json")):
        pytest.skip("no BN254 golden file")
    out = _run(["tests/test_gpu_parity_pins.py"], extra=["-k", "golden_xl_cfg"])
    assert " passed" in out
