import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def gpu():
    """Initialised libmarlin_hip on device 0.  Fails loudly (no skip, no CPU fallback)
    when the HIP library or the GPU is missing."""
    import marlin_amd
    marlin_amd.init(0)
    if os.environ.get("MARLIN_TEST_POISON"):          # tests/test_gpu_poisoned_allocations.py: every allocation starts as 0xA5 garbage
        from marlin_amd import _lib
        _lib.check(_lib.load().mh_debug_poison_scratch(1), "mh_debug_poison_scratch")
    yield marlin_amd
