import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "cugraph-gnn_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def hiplib():
    """The built C-ABI library; building it (hipcc cross-compiles without a GPU) if needed."""
    import __graft_entry__ as g
    from wholegraph_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        g.build()
    return _lib.lib()
