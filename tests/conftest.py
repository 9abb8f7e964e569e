import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def lib():
    """The in-tree libunerf.so, built on demand (hipcc cross-compiles without a GPU)."""
    from uncertainty_nerf_gs_amd import lib as L
    L.build_library()
    return L


@pytest.fixture(scope="session")
def dev(lib):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no HIP device in this container (GPU tests run on the MI355X box)")
    lib.require_gpu()
    return torch.device("cuda:0")


def golden(name):
    import numpy as np
    return np.load(os.path.join(GOLDEN, name))
