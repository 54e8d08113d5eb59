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
    yield marlin_amd
