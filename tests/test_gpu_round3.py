"""-m gpu tests added in round 3: Dynamics._ticks, the NaN / inf guard counter, the faithful
checkpoint round trip of pending Philox perturbations, served (persistent) stepping, the per-component
pure-relative parity report and a real RCCL collective on one GPU."""
import os

import numpy as np
import pytest

from conftest import load_cases
from gpu_util import (MODE_TOL, VecOracle, assert_state_close, assert_step_close, have_gpu, make_pair,
                      scaled_err, step_both, to_np)

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not have_gpu(), reason="needs a HIP device")]

DYN = load_cases("dynamics_traces.npz")
ENV = load_cases("env_traces.npz", "variant_traces.npz")
HOVER = float(np.load(os.path.join(os.path.dirname(__file__), "golden", "meta.npz"))["hover_motor"])


# ---------------------------------------------------------------------------------------
# Dynamics._ticks / getTime() (dynamics/__init__.py:98, :197, :219-221)
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("mode", ["float32", "float64"])
@pytest.mark.parametrize("fps", [100, 1000])
def test_ticks_follow_the_golden_dynamics_traces(fps, mode):
    """The reference's own tick counter, recorded after every setMotors call of the D-series (take-off,
    crash, soft landing: a ground-contact freeze does not tick), against cs_set_motors + track_time."""
    import torch
    cs = [c for c in DYN.names() if int(DYN[c]["fps"]) == fps and (mode == "float64" or c != "D12_full_range")]
    n = len(cs)
    T = max(len(DYN[c]["status"]) for c in cs)
    env, _ = make_pair("lander3d", n, mode, frames_per_second=fps, track_time=True)
    x0 = np.stack([DYN[c]["x0"] for c in cs], axis=1)
    status0 = np.array([int(DYN[c]["status0"]) for c in cs], np.uint8)
    force = np.stack([DYN[c]["force"][:3] for c in cs], axis=1)
    flags = np.array([5 if np.any(DYN[c]["force"]) else 0 for c in cs], np.uint8)
    motors = np.zeros((T, n, 4), dtype=np.float32)
    for i, c in enumerate(cs):
        motors[:len(DYN[c]["motors"]), i] = DYN[c]["motors"]
    env.set_state(x=x0, status=status0, force=force, flags=flags, steps=np.ones(n, np.int32),
                  ticks=np.zeros(n, np.int32))
    saw_freeze = False
    for t in range(T):
        env.set_motors(torch.from_numpy(motors[t]).to(env.device))
        if t % 25 and t != T - 1:
            continue
        st = env.get_state()
        dev = to_np(env.state_tensors()["ticks"])
        for i, c in enumerate(cs):
            g = DYN[c]
            if t < len(g["status"]) and st["status"][i] == g["status"][t]:
                assert st["ticks"][i] == g["ticks"][t] == dev[i], (c, t, st["ticks"][i], g["ticks"][t])
                saw_freeze |= int(g["ticks"][t]) != t + 1
    assert saw_freeze or fps == 1000          # the fps-100 set holds ground contacts
    # getTime() = ticks * dt
    assert np.allclose(to_np(env.get_time()), env.get_state()["ticks"] / float(fps))
    env.close()


@pytest.mark.parametrize("substeps", [1, 4])
def test_ticks_through_env_steps_match_the_oracle(substeps):
    """_Task.step skips the physics of a LANDED env and a contact freeze does not tick: ticks != steps.
    Every kernel that advances an env (one step, K steps, auto-reset) keeps the counter."""
    import torch
    n = 640
    env, orc = make_pair("lander3d", n, "float32", autoreset="next_step", substeps=substeps, seed=3,
                         track_time=True, initial_altitude=0.6)
    env.reset()
    orc.reset()
    rng = np.random.default_rng(4)
    for t in range(160):
        a = (HOVER * (0.97 + 0.04 * rng.random((n, 4)))).astype(np.float32)      # settle, land, some crash
        if t % 3 == 0:
            got = env.step_many(torch.from_numpy(a[None]).to(env.device))
            want = orc.step(a.astype(np.float64))
            assert np.array_equal(to_np(got[2])[0], want[2])
        else:
            got, want, _ = step_both(env, orc, a)
            assert np.array_equal(got[2].astype(bool), want[2])
    st = env.get_state()
    assert np.array_equal(st["ticks"], orc.ticks)
    assert np.array_equal(st["steps"], orc.steps)
    assert (st["ticks"] != substeps * (st["steps"] - 1)).any()      # the two counters really differ
    env.close()


def test_ticks_are_reported_as_minus_one_without_track_time():
    env, _ = make_pair("lander3d", 100, "float32")
    env.reset()
    assert (to_np(env.state_tensors()["ticks"]) == -1).all()
    with pytest.raises(RuntimeError):
        env.get_time()
    env.close()


# ---------------------------------------------------------------------------------------
# NaN / inf guard counter (upstream propagates silently, task.py:133)
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("mode", ["float32", "float64"])
def test_nonfinite_guard_counter(mode):
    import torch
    n = 1000
    env, _ = make_pair("hover3d", n, mode)
    env.reset()
    names = env.STATS_NAMES
    assert names[6] == "nonfinite" and float(to_np(env.batch_stats())[6]) == 0.0
    a = np.full((n, 4), HOVER, np.float32)
    bad = np.zeros(n, bool)
    bad[[3, 64, 65, 700, 999]] = True
    a[bad, 1] = np.nan                       # a NaN action reaches the state through the motor model
    a[500, 2] = np.inf                       # clipped to 1 by np.clip: stays finite
    env.step(torch.from_numpy(a).to(env.device))
    stats = to_np(env.batch_stats())
    assert stats[6] == bad.sum() and stats[0] == n
    x = env.get_state()["x"]
    assert np.array_equal(~np.isfinite(x).all(axis=0), bad)
    env.step(torch.from_numpy(np.full((n, 4), HOVER, np.float32)).to(env.device))
    assert to_np(env.batch_stats())[6] == bad.sum()      # they stay non-finite, as upstream's would
    env.close()


# ---------------------------------------------------------------------------------------
# checkpoint round trip: pending Philox perturbations stay Philox
# ---------------------------------------------------------------------------------------
def test_state_round_trip_keeps_philox_perturbations_on_the_seed():
    import torch
    n = 300
    a, _ = make_pair("lander3d", n, "float32", seed=11)
    b, _ = make_pair("lander3d", n, "float32", seed=11)
    explicit = np.zeros((3, n), np.float32)
    explicit[:, ::7] = 5.0
    for e in (a, b):
        e.reset()
        e.set_perturbation(explicit, mask=(np.arange(n) % 7 == 0))      # some envs hold an installed force
    st = a.get_state()
    assert np.array_equal((st["flags"] & 4) != 0, np.arange(n) % 7 == 0) and (st["flags"] & 1).all()
    a.set_state(**st)                                                    # restore what was saved
    st2 = a.get_state()
    for k in st:
        assert np.array_equal(st[k], st2[k], equal_nan=True), k
    # a re-seed moves the pending Philox draws of BOTH envs alike; the installed forces stay
    a.seed(99)
    b.seed(99)
    fa, fb = a.get_state()["force"], b.get_state()["force"]
    assert np.array_equal(fa, fb) and not np.array_equal(fa, st["force"])
    assert np.array_equal(fa[:, ::7], explicit[:, ::7].astype(np.float64))
    act = torch.full((n, 4), HOVER, dtype=torch.float32, device=a.device)
    oa, ob = a.step(act)[0], b.step(act)[0]
    assert torch.equal(oa, ob)
    a.close()
    b.close()
