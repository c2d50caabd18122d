#!/usr/bin/env python3
"""GPU box: a one-off sweep of the combinatorial parity fuzz (tests/test_gpu_fuzz.py) over more seeds than the
suite runs (the suite: 0..63).
  python tools/fuzz_sweep.py [first] [last]      # default 64 400
Prints one line per failing seed (with the assertion) and the count; exit code 0 either way (a report, not a gate)."""
import os
import sys
import tempfile
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_fuzz as F  # noqa: E402


class _Tmp:
    def __init__(self):
        self.d = tempfile.mkdtemp(prefix="fuzz_sweep_")
        os.chmod(self.d, 0o700)

    def getbasetemp(self):
        import pathlib
        return pathlib.Path(self.d)


first, last = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (64, 400)
tmp, bad = _Tmp(), []
fn = F.test_random_configuration_and_stepping_forms_vs_oracle
fn = getattr(fn, "__wrapped__", fn)
for seed in range(first, last):
    try:
        fn(seed, tmp)
    except Exception as e:      # noqa: BLE001
        bad.append(seed)
        msg = str(e) if os.environ.get("FUZZ_FULL") else str(e).splitlines()[0][:400]
        print("seed %d FAILED: %s" % (seed, msg), flush=True)
        if os.environ.get("FUZZ_TRACE"):
            traceback.print_exc()
print("fuzz sweep seeds %d..%d: %d passed, %d failed %s" % (first, last - 1, last - first - len(bad), len(bad), bad))
