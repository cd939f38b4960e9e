import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def _built_libraries():
    """A fresh checkout has no binaries (they are git-ignored): build the
    product library (hipcc cross-compiles without a GPU) and the oracle once
    per session. The product itself never builds on demand -- it fails
    loudly when liblbmi.so is missing."""
    from ludwig_amd import lib as L
    if not os.path.exists(L.LIB_PATH):
        L.build()
    from oracle import lb_oracle as lbo
    lbo.lib()
    yield
