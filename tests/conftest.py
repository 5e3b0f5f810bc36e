import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    """Golden vectors written by tests/golden/make_golden.py from the reference's own functions."""
    with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as f:
        return {k: f[k] for k in f.files}


@pytest.fixture
def golden():
    return load_golden


class _LvaeEnv:
    """LVAE_* development switches are read once per process by liblvae.so (lvae_dev_reload_env re-reads them):
    set / unset one for the current test AND have the library see it."""

    def __init__(self, monkeypatch):
        self._mp = monkeypatch

    def set(self, name, value):
        from lvae import _native as N
        self._mp.setenv(name, str(value))
        N.reload_env()

    def unset(self, name):
        from lvae import _native as N
        self._mp.delenv(name, raising=False)
        N.reload_env()


@pytest.fixture(autouse=True)
def _lvae_env_restored():
    """Runs its teardown AFTER monkeypatch has restored the environment (autouse fixtures are set up first, torn down
    last): the library re-reads the restored switches, so no test leaves a stale one behind."""
    yield
    from lvae import _native as N
    if N._lib is not None:
        N.reload_env()


@pytest.fixture
def lvae_env(monkeypatch):
    return _LvaeEnv(monkeypatch)
