import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def hiplib():
    """The in-tree HIP library; built on demand (hipcc cross-compiles without a GPU)."""
    from lpslam_amd import _build, hip
    _build.hip_library()
    hip.load()
    return hip


def golden(name):
    import numpy as np
    return np.load(os.path.join(GOLDEN, name))
