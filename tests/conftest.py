import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def pytest_report_header(config):
    """Says in the first lines of every run whether the torchvision cross-check (tests/test_tv_crosscheck.py) can run here."""
    try:
        import torchvision
        return "torchvision cross-check: RUNS (torchvision %s installed)" % torchvision.__version__
    except Exception as e:                                   # noqa: BLE001 -- any import failure means "not available"
        return "torchvision cross-check: SKIPPED, torchvision is not importable here (%s); nms / RoIPool / RoIAlign / AnchorGenerator stay 'parity unpinned'" % type(e).__name__


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return load
