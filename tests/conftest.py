import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_PARENT = os.path.join(ROOT, "3d-object-detection_amd")
for p in (ROOT, PKG_PARENT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a ROCm GPU (run on the MI355X box)")


@pytest.fixture(scope="session")
def device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no ROCm device")
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _seeded_global_generators():
    """Every test starts from the same state of the process-wide generators (torch's default
    CPU/GPU generators, numpy's legacy one): a test that draws from them without a generator
    of its own gives the same numbers in every run and in any selection of tests."""
    import numpy as np
    import torch
    torch.manual_seed(20240611)
    np.random.seed(20240611)
    yield
