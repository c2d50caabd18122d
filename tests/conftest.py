"""pytest configuration: registers the `gpu` marker and shared fixture loaders."""
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


class Cases:
    """Read-only view of one golden .npz: cases[name][field]."""

    def __init__(self, path):
        z = np.load(path)
        self._d = {}
        for k in z.files:
            c, f = k.split("/", 1)
            self._d.setdefault(c, {})[f] = z[k]

    def names(self):
        return sorted(self._d)

    def __getitem__(self, name):
        return self._d[name]


_cache = {}


def load_cases(fname):
    if fname not in _cache:
        _cache[fname] = Cases(os.path.join(GOLDEN, fname))
    return _cache[fname]


@pytest.fixture(scope="session")
def dyn_cases():
    return load_cases("dynamics_traces.npz")


@pytest.fixture(scope="session")
def env_cases():
    return load_cases("env_traces.npz")
