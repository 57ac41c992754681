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


@pytest.fixture(scope="session")
def oracle():
    from oracle import binding
    return binding


class _HipOptions(object):
    """the library's switches for the length of a test (carmel_hip_set_option; until round 5 these were CARMEL_HIP_* environment
    variables the library read itself).  The variable is exported as well: child processes -- the front ends, bench.py, the
    multi-rank workers -- translate their environment into options themselves (csrc/host/env_options.hpp)."""

    def __init__(self, monkeypatch):
        self.mp, self.saved = monkeypatch, {}

    @staticmethod
    def _env(key):
        return "CARMEL_TIMING" if key == "timing" else "CARMEL_HIP_" + key.upper()

    def set(self, key, value):
        import carmel_amd
        self.saved.setdefault(key, carmel_amd.get_option(key))
        carmel_amd.set_option(key, None if value is None else str(value))
        if value is None:
            self.mp.delenv(self._env(key), raising=False)
        else:
            self.mp.setenv(self._env(key), str(value))

    def unset(self, key):
        self.set(key, None)

    def undo(self):
        import carmel_amd
        for k, v in self.saved.items():
            carmel_amd.set_option(k, v)
        self.saved = {}


@pytest.fixture
def hipopt(monkeypatch):
    h = _HipOptions(monkeypatch)
    yield h
    h.undo()
