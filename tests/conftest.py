import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _build_everything():
    """Build the product .so and the oracle once per session if they are missing (hipcc
    cross-compiles gfx950 without a GPU).  On the GPU box the prebuilt files travel with the repo."""
    import _libs as L
    need = not os.path.exists(L.PRODUCT_SO) or not os.path.exists(L.ORACLE_SO)
    if need:
        sys.path.insert(0, L.ROOT)
        import __graft_entry__ as g
        g.build()
    yield


def pytest_collection_modifyitems(config, items):
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU in this container (GPU tests run via gpurun / -m gpu)")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)
