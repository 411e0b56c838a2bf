import os
import sys

import pytest

try:        # not needed for the load order any more (_native.load() keeps the process on one HIP runtime either way); imported
    import torch  # noqa: F401  # here so that the 1-2 minutes of a first `import torch` on a fresh box are not billed to a test
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
