import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def kats():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "reference_kats.json")) as f:
        return json.load(f)


@pytest.fixture(autouse=True)
def _switches_follow_the_environment(monkeypatch):
    """The library reads its LDPC_HIP_* switches once (kernels.hpp Switches); tests flip them with monkeypatch, so
    every setenv / delenv is followed by a reload, and each test starts from the environment as it is."""
    def reload():
        try:
            import ldpc_amd
            ldpc_amd.reload_switches()
        except ImportError:
            pass

    set0, del0 = monkeypatch.setenv, monkeypatch.delenv

    def setenv(name, value, prepend=None):
        set0(name, value, prepend)
        reload()

    def delenv(name, raising=True):
        del0(name, raising)
        reload()

    monkeypatch.setenv, monkeypatch.delenv = setenv, delenv
    reload()
    yield
