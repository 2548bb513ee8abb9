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
    """No test's verdict depends on the process-wide generators (torch's default CPU/GPU
    generators, numpy's legacy one): every test draws from generators it seeds itself.  They
    are still put into a known state before each test so that a run is reproducible; the
    state comes from FPN_TEST_SEED, and the suite passes for any value of it."""
    import numpy as np
    import torch
    seed = int(os.environ.get("FPN_TEST_SEED", "20240611"))
    torch.manual_seed(seed)
    np.random.seed(seed % (1 << 32))
    yield
