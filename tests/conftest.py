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


@pytest.fixture
def monkeypatch(monkeypatch):
    """The stock fixture, plus: the package and the C library read their HDRSKY_* variables once (hooks.py, csrc/hooks.h),
    so every setenv / delenv of a test is followed by a reload - and one more when the test's changes are undone.  Tuning
    hooks additionally need HDRSKY_EXPERIMENTS=1, which a test sets like any other variable."""
    def reload():
        pkg("hooks").reload()
    set0, del0 = monkeypatch.setenv, monkeypatch.delenv
    def setenv(name, value, prepend=None):
        set0(name, value, prepend); reload()
    def delenv(name, raising=True):
        del0(name, raising); reload()
    monkeypatch.setenv, monkeypatch.delenv = setenv, delenv
    yield monkeypatch
    monkeypatch.undo()
    reload()
