import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (checker only).  Built on demand with gcc."""
    from oracle import xm_oracle as xo
    xo.lib()
    return xo


@pytest.fixture(scope="session")
def xmamd():
    """ctypes binding of the product library (xm-code_amd/lib/libxm_amd.so)."""
    p = os.path.join(ROOT, "xm-code_amd")
    if p not in sys.path:
        sys.path.insert(0, p)
    import xmamd as m
    m.lib()
    return m
