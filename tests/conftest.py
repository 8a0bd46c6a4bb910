import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    import __graft_entry__ as ge
    o = ge.load_oracle()
    o.lib()
    return o


@pytest.fixture(scope="session")
def pkg():
    import __graft_entry__ as ge
    p = ge.load_package()
    return p


@pytest.fixture(scope="session")
def gpu(pkg):
    """The product library on a real device; fails loudly (no fallback) when either is missing."""
    import torch
    assert torch.cuda.is_available(), "-m gpu tests need a GPU"
    pkg.lib()
    pkg.device_check()
    return pkg
