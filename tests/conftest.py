import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
PKG = "hdr-map-reconstruction-from-a-single-ldr-sky-panoramic-image-for-outdoor-illumination-estimation_amd"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def pkg(sub=None):
    return importlib.import_module(PKG + ("." + sub if sub else ""))


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _collect_between_tests():
    """Objects that own hipGraphs / events and sit in reference cycles (a Trainer and its plan's closures) are finalised HERE,
    between tests, not by a collection that starts in the middle of the next test's graph capture (torch.cuda.graph does not
    collect on entry any more; hipGraphExecDestroy on a capturing thread aborts the process)."""
    yield
    import gc
    gc.collect()
