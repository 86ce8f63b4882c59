import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lib():
    import oracle
    return oracle.lib()


@pytest.fixture(scope="session")
def hip_lib():
    """The product library; GPU tests must run on it -- no fallback."""
    from sparkl_amd import _lib
    return _lib.lib()
