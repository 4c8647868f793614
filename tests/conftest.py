"""pytest wiring: `gpu` marker, import paths, and a session-scoped build of libspx.so + the C oracle.

  python -m pytest tests/ -x -q -m "not gpu"   # oracle vs golden vectors, host logic, ABI exports (no GPU needed)
  python -m pytest tests/ -x -q -m gpu         # HIP-vs-oracle parity, through the C ABI, on an MI355X
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "tsm-det-pointcloud-_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Compile the HIP library (cross-compiles without a GPU) and the oracle once per session."""
    import __graft_entry__ as ge
    ge.build(verbose=False)
    yield


@pytest.fixture(scope="session")
def orc():
    from oracle import oracle
    oracle.lib()
    return oracle


def gpu_available():
    import torch
    return torch.cuda.is_available()
