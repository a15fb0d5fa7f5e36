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


@pytest.fixture(autouse=True)
def _poison_device_cache(request):
    """WGAMD_TEST_POISON=1: before every GPU test, fill a few hundred MB of the caching allocator's free blocks with a
    non-zero pattern, so that code which reads memory it never wrote (capacity slack, missing clears) sees garbage instead of
    the zeros fresh pages happen to hold.  Off by default (costs time); used for hardening runs."""
    if os.environ.get("WGAMD_TEST_POISON") == "1" and request.node.get_closest_marker("gpu") is not None:
        import torch
        if torch.cuda.is_available():
            blocks = [torch.empty(n, dtype=torch.int32, device="cuda").fill_(0x7F7F7F7F) for n in (1 << 26, 1 << 24, 1 << 22, 1 << 20)]
            blocks += [torch.empty(1 << 16, dtype=torch.int32, device="cuda").fill_(-3) for _ in range(64)]
            torch.cuda.synchronize()
            del blocks
    yield


def pytest_collection_modifyitems(config, items):
    """WGAMD_TEST_SHUFFLE=<seed>: run the collected tests in a seeded random order (hardening runs: no test may depend
    on allocator state, cached walks or communicators left behind by another)."""
    seed = os.environ.get("WGAMD_TEST_SHUFFLE")
    if seed:
        import random
        random.Random(int(seed)).shuffle(items)
