import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Native artefacts: libvdl2gpu.so (hipcc cross-compiles without a GPU) + liboracle.so."""
    import __graft_entry__ as g
    g.build()
    return True


@pytest.fixture(scope="session")
def oracle(built):
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def have_ref(oracle):
    return oracle.have_ref()
