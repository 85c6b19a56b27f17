import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a ROCm GPU (run with -m gpu on the MI355X box)")


@pytest.fixture(scope="session")
def lib():
    """The built C-ABI library (built on demand; hipcc cross-compiles without a GPU)."""
    from spgnn_amd.csrc import build as _b  # noqa
    _b.build(verbose=False)
    from spgnn_amd import _capi
    return _capi.load()
