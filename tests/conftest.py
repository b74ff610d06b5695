import os
import sys

import pytest

# torch first: it bundles its own ROCm runtime libraries, the engine's .so links the system ones, and the two
# share one copy per SONAME in a process -- whichever is loaded first.  With the system copy loaded first (a test
# that creates an engine before anything imported torch) torch later finds "No HIP GPUs"; the other order works
# (bench.py imports torch first, too).
try:
    import torch  # noqa: F401
except Exception:                                   # pragma: no cover
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
