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
    from oracle import orc_bind
    orc_bind.build()
    return orc_bind.lib()


@pytest.fixture(autouse=True)
def _seed_libc_rand():
    """The reference's builders (and this build's, for source compatibility) draw Xavier weights from libc rand().
    Seed it per test so that every test sees the same parameters in every run and in any test order -- graphs
    with batch-norm over few samples are sensitive enough for an unlucky draw to move a 1e-4 comparison."""
    import ctypes
    ctypes.CDLL(None).srand(20240607)
    yield
