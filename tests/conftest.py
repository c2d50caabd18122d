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

    def __init__(self, *paths):
        self._d = {}
        for path in paths:
            z = np.load(path)
            for k in z.files:
                c, f = k.split("/", 1)
                self._d.setdefault(c, {})[f] = z[k]

    def names(self):
        return sorted(self._d)

    def __getitem__(self, name):
        return self._d[name]


_cache = {}


def load_cases(*fnames):
    """Cases of one golden file, or of several merged (case names are unique across files)."""
    if fnames not in _cache:
        _cache[fnames] = Cases(*[os.path.join(GOLDEN, f) for f in fnames])
    return _cache[fnames]


@pytest.fixture(scope="session")
def dyn_cases():
    return load_cases("dynamics_traces.npz")


@pytest.fixture(scope="session")
def env_cases():
    return load_cases("env_traces.npz")
