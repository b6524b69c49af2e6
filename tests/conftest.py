import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def _reload_library_switches():
    """The library snapshots its GTARS_* switches at first use (gtars_amd/csrc/common.h: cfg_get): a test that changes one asks
    for a new snapshot."""
    m = sys.modules.get("gtars_amd")
    if m is not None:
        m.reload_env()


@pytest.fixture
def monkeypatch(monkeypatch):
    """pytest's monkeypatch, with setenv / delenv followed by a new snapshot of the library's switches (and again when the
    test's changes are undone)."""
    setenv, delenv = monkeypatch.setenv, monkeypatch.delenv

    def setenv_and_reload(name, value, prepend=None):
        setenv(name, value, prepend)
        _reload_library_switches()

    def delenv_and_reload(name, raising=True):
        delenv(name, raising)
        _reload_library_switches()

    monkeypatch.setenv = setenv_and_reload
    monkeypatch.delenv = delenv_and_reload
    yield monkeypatch
    monkeypatch.undo()
    _reload_library_switches()
