import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    import __graft_entry__ as g
    return g.load_pkg()


@pytest.fixture(scope="session")
def oracle():
    import __graft_entry__ as g
    return g.load_oracle()


@pytest.fixture(scope="session")
def fir(pkg):
    return pkg.if_fir


@pytest.fixture(scope="session")
def gpu_ok():
    """GPU tests must not pass on a silent fallback: fail loudly when no device is visible."""
    import torch
    assert torch.cuda.is_available(), "gpu-marked test running without a visible HIP device"
    return True
