import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def hip():
    """The C-ABI binding; GPU tests go through it (and fail loudly if the .so is absent)."""
    import torch
    assert torch.cuda.is_available(), "gpu-marked test run without a GPU"
    from rslo_amd import capi
    capi.lib()
    return capi
