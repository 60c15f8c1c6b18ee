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
    """Build both libraries if they are missing (no-op on the GPU box: the .so files travel with the snapshot)."""
    need = [os.path.join(ROOT, "pathtracing_amd", "libptrt.so"), os.path.join(ROOT, "oracle", "libpt_oracle.so")]
    if not all(os.path.exists(p) for p in need):
        import __graft_entry__ as g
        g.build()


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
