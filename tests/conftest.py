import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def solve_mod():
    """The ctypes binding of the HIP library; GPU tests go through the C-ABI only."""
    from epsilon_amd import _solve
    if not os.path.exists(_solve.LIB_PATH):
        from epsilon_amd import build
        build.build()
    assert _solve.device_count() > 0, "GPU test on a box without a HIP device"
    return _solve
