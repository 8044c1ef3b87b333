import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(autouse=True)
def _kernel_selection_follows_the_environment(monkeypatch):
    """The product reads its kernel selection ONCE (ops.configure(), called by prepare()), not per call.  Tests that switch a variant
    with monkeypatch.setenv("MQ_<NAME>", ...) between two calls in one process get the table re-read right after the setenv; every
    test starts from the defaults of the (clean) environment."""
    from mq_det_amd import ops
    ops.configure()
    orig_set, orig_del = monkeypatch.setenv, monkeypatch.delenv

    def setenv(name, value, *a, **k):
        orig_set(name, value, *a, **k)
        if name.startswith("MQ_"):
            ops.configure()

    def delenv(name, *a, **k):
        orig_del(name, *a, **k)
        if name.startswith("MQ_"):
            ops.configure()
    monkeypatch.setenv, monkeypatch.delenv = setenv, delenv
    yield
    monkeypatch.undo()
    ops.configure()
