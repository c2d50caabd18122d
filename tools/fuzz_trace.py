#!/usr/bin/env python3
"""GPU box: fly ONE case of the combinatorial parity fuzz (tests/test_gpu_fuzz.py) without stopping at the first
assertion and print how the differences between device and oracle develop -- per step the largest observation
difference (scaled as the suite scales it) and the largest reward difference with the env's |shaping| beside it, per
stretch the state difference and the env that carries it.  Discrete outputs (flags, status, counters) still assert.

    python3 tools/fuzz_trace.py SEED

Used to tell chaos (a one-unit difference of the stored format that grows step by step on ONE lane flying at hundreds
of m/s or rad/s) from a defect (a difference that appears at once, on many lanes, or in a discrete output)."""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gpu_util as U          # noqa: E402
import test_gpu_fuzz as F     # noqa: E402

seed = int(sys.argv[1])
log = []


def step_close(got, want, x_tol, r_abs=5e-5, r_rel=1e-5, ctx="", r_unit=0.0, shaping=None):
    obs, r, term, trunc = got
    wobs, wr, wterm, wtrunc = want
    assert np.array_equal(term.astype(bool), wterm) and np.array_equal(trunc.astype(bool), wtrunc), "flags " + ctx
    o64, w64 = np.asarray(obs, np.float64), np.asarray(wobs, np.float64)
    with np.errstate(invalid="ignore"):
        eo = np.abs(o64 - w64) / np.maximum(np.abs(w64), U.SCALE)
    eo = np.where(np.isfinite(eo), eo, 0.0)
    dr = np.abs(r.astype(np.float64) - wr)
    dr = np.where(np.isfinite(dr), dr, 0.0)
    over = dr - (r_abs + r_rel * np.abs(wr))          # beyond the suite's fixed part (float32 rounding of a large reward)
    j, i = int(np.argmax(over)), int(np.argmax(eo.max(axis=1)))
    sh = float(shaping[j]) if shaping is not None else float("nan")
    what = ctx.split(" form ")[-1]
    print("%-16s obs %.2e (env %4d)   reward %.2e (fixed limit %.2e) at env %4d, r = %+.4g: |shaping| %.4g -> %.1e of it; "
          "largest observed component there %.4g" % (what, eo.max(), i, dr[j], dr[j] - over[j], j, wr[j], sh,
                                                     dr[j] / max(sh, 1.0), float(np.max(np.abs(w64[j])))))


def state_close(env, orc, x_tol, ctx=""):
    s = env.get_state()
    assert np.array_equal(s["status"], orc.status) and np.array_equal(s["steps"], orc.steps), "discrete state " + ctx
    x, w = np.asarray(s["x"], np.float64), np.asarray(orc.x, np.float64)
    with np.errstate(invalid="ignore"):
        e = np.abs(x - w) / np.maximum(np.abs(w), U.SCALE)
    e = np.where(np.isfinite(e), e, 0.0)
    lane_axis = int(np.argmax(np.array(e.shape) == orc.status.shape[0])) if e.ndim == 2 else 0
    per_env = e.max(axis=1 - lane_axis) if e.ndim == 2 else e
    j = int(np.argmax(per_env))
    print("   == state after %s: %.2e (tolerance %.1e) on env %d; envs above half the tolerance: %d of %d"
          % (ctx.split(" form ")[-1], per_env[j], x_tol, j, int(np.sum(per_env > 0.5 * x_tol)), per_env.size))
    return float(per_env[j])


F.assert_step_close, F.assert_state_close = step_close, state_close


class _Tmp:
    def __init__(self):
        self.d = tempfile.mkdtemp(prefix="fuzz_trace_")

    def getbasetemp(self):
        import pathlib
        return pathlib.Path(self.d)


cfg, kw, law, _ = F.draw_case(seed)
print("# seed %d: %r %r law %s" % (seed, cfg, kw, law))
F.run_case(seed, _Tmp())
