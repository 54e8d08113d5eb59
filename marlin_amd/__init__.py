"""marlin_amd -- MI355X (gfx950) implementation of the Marlin prover hot path.

Python is plumbing only: it binds the C ABI of libmarlin_hip.so
(include/marlin_hip.h) so tests and bench.py can drive it, mirroring the names
of the reference interfaces each call replaces:

  ntt / intt           <- ark_poly GeneralEvaluationDomain::{fft, ifft}
                          (/root/reference src/ahp/prover.rs:326,350,...)
  Bases + msm          <- ark_ec VariableBaseMSM::multi_scalar_mul reached through
                          PC::commit / PC::open_combinations (src/lib.rs:172-292)

Field elements travel as numpy uint64 arrays of shape (n, 4) (Fr) / (n, 6) (Fq)
in arkworks' in-memory layout (little-endian limbs, Montgomery form).
"""
from ._lib import MarlinHipError, load, check, LIB_PATH  # noqa: F401
from .api import (init, shutdown, device_info, ntt, intt, coset_ntt, ntt_dev, Bases, G2Bases, g2_msm, msm, msm_dev, msm_batch_dev, msm_batch_sharded_dev,  # noqa: F401
                  DeviceBuffer, g1_to_affine, msm_path_counts, prof_enable, prof_reset, prof_get, synchronize)
