import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "what-matters-for-meta-learning_amd")
for p in (PKG, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hostsim():
    """TEST-ONLY host build of the library's index arithmetic (tests/hostsim/)."""
    from mlhot.binding import MlhotLib
    from mlhot.build import build_hostsim
    return MlhotLib(build_hostsim(os.path.join(ROOT, "tests", "hostsim")))


@pytest.fixture(scope="session")
def gpulib():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test started without a GPU")
    import mlhot
    mlhot.build_product()
    return mlhot.lib()
