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


def pytest_collection_modifyitems(config, items):
    """A GPU test that hangs must fail by itself, not take the whole run with it: every gpu test gets a 700 s
    limit when pytest-timeout is there (the slowest, the RCCL children with one retry, stay under 11 minutes
    even on a cold box; a normal test takes well under a second)."""
    if not config.pluginmanager.hasplugin("timeout"):
        return
    for item in items:
        if item.get_closest_marker("gpu") is not None and item.get_closest_marker("timeout") is None:
            item.add_marker(pytest.mark.timeout(700))


def pytest_runtest_logstart(nodeid, location):
    """The test now running, for a run that never comes back (gpurun merges gpurun_out/ also after a timeout)."""
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "current_test.txt"), "w") as f:
            f.write(nodeid + "\n")
    except OSError:
        pass


def pytest_runtest_logreport(report):
    """Any gpu test that takes more than 10 s leaves a line (and what it printed) in gpurun_out/slow_tests.txt:
    a run that was slow or was cut off can be explained afterwards."""
    if report.when != "call" or report.duration < 10.0 or "gpu" not in report.keywords:
        return
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "slow_tests.txt"), "a") as f:
            f.write("%8.1f s  %s  %s\n%s\n" % (report.duration, report.outcome, report.nodeid, report.capstdout[-8000:]))
    except OSError:
        pass


def pytest_sessionstart(session):
    """A fresh checkout has no built artefacts (they are git-ignored): build the library and the C++
    host programs once, exactly as __graft_entry__.build() does, when hipcc is available.  (hipcc
    cross-compiles for gfx950 without a GPU; on the GPU box the built files arrive with the snapshot.)"""
    import shutil
    import subprocess
    lib = os.path.join(ROOT, "gym_copter_amd", "libcopterstep.so")
    host = os.path.join(ROOT, "tests", "host", "rollout_policy_host")    # the last target of `make all`
    hipcc = shutil.which("hipcc") or ("/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else None)
    if (not os.path.exists(lib) or not os.path.exists(host)) and hipcc:
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "gym_copter_amd", "csrc"), "all",
                               "PYTHON=" + sys.executable], stdout=subprocess.DEVNULL)


def pytest_sessionfinish(session, exitstatus):
    """Leave the GPU quiet before the interpreter goes down: envs a test left open are closed, hipGraphs and streams
    the tests dropped are collected, the device is idle.  Destroying device objects from the interpreter's shutdown
    (or from the runtime's own exit handlers, in whatever order) has ended a run in which every test had passed with
    SIGSEGV at exit on one box -- which a driver reads as a failed run."""
    torch = sys.modules.get("torch")
    vecenv = sys.modules.get("gym_copter_amd.vecenv")
    if vecenv is not None:
        vecenv._close_open_envs()
    if torch is not None and torch.cuda.is_initialized():
        import gc
        gc.collect()
        try:
            torch.cuda.synchronize()
        except Exception:
            pass


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
