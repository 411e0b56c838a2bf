import os
import sys

import pytest

try:        # torch brings its own HIP runtime: it has to be loaded before libreloc_hip.so pulls in /opt/rocm's, or torch later
    import torch  # noqa: F401  # reports "No HIP GPUs are available" (only the multi-GPU exchange and bench.py use torch)
except ImportError:
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure).  Built on demand with gcc."""
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def engine():
    """One HIP engine for the whole GPU test session (fails loudly without a GPU)."""
    from nclt_slam_project_amd.engine import Engine
    e = Engine(device=0, max_w=1280, max_h=720, max_feat=8192)
    yield e
    e.close()
