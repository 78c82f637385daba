import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a HIP device (run on the MI355X box with -m gpu)")


@pytest.fixture(scope="session")
def ctx():
    """One vo_ctx on device 0 for the whole GPU session (fails loudly without a GPU: no fallback)."""
    from visual_odometry_amd import _lib
    return _lib.Context.default()


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    path = os.path.join(ROOT, "tests", "golden", "golden.npz")
    return np.load(path, allow_pickle=False)
