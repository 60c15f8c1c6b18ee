import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))  # tests are allowed to use the oracle (the checker)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Make sure both libraries are current with their sources (a stale checker once hid behind an old .so)."""
    import __graft_entry__ as g
    g.build()  # `make` is incremental: a no-op when the .so files are newer than their sources


@pytest.fixture(scope="session")
def P():
    import pathtracing_amd
    return pathtracing_amd


@pytest.fixture(scope="session")
def pto():
    import pto as _pto
    return _pto


@pytest.fixture(scope="session")
def renderer(P):
    """One context for the whole GPU session (tests run in one process)."""
    r = P.Renderer(P.Window(256, 256))
    r.Init()
    yield r
    r.Dispose()
