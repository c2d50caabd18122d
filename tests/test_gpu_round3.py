"""-m gpu tests added in round 3: Dynamics._ticks, the NaN / inf guard counter, the faithful
checkpoint round trip of pending Philox perturbations, served (persistent) stepping, the per-component
pure-relative parity report and a real RCCL collective on one GPU."""
import os

import numpy as np
import pytest

from conftest import load_cases
from gpu_util import (MODE_TOL, assert_state_close, assert_step_close, have_gpu, make_pair,
                      step_both, to_np)

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
    for t in range(240):
        a = (HOVER * (0.93 + 0.05 * rng.random((n, 1))) * np.ones((1, 4))).astype(np.float32)   # sink: soft landings and crashes
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


# ---------------------------------------------------------------------------------------
# served stepping: one persistent env kernel per session (cs_serve_*, include/copterstep_serve.h)
# ---------------------------------------------------------------------------------------
def _twin(task, n, mode, **kw):
    import gym_copter_amd
    mk = lambda: gym_copter_amd.CopterVecEnv(task=task, num_envs=n, state_dtype=mode, seed=21, **kw)
    a, b = mk(), mk()
    a.reset()
    b.reset()
    return a, b


def _assert_same_state(a, b):
    sa, sb = a.get_state(), b.get_state()
    for k in sa:
        assert np.array_equal(sa[k], sb[k], equal_nan=True), k


@pytest.mark.parametrize("task,mode,kw", [
    ("lander3d", "float32", dict(autoreset_mode="next_step")),
    ("hover3d", "float32", dict(autoreset_mode="next_step")),
    ("lander3d", "float64", dict(autoreset_mode="same_step", episode_stats=True, track_time=True)),
    ("lander3d", "float32_rn", dict(autoreset_mode="disabled", substeps=3)),
    ("lander2d", "float32", dict(autoreset_mode="next_step")),
    ("hover1d", "float32", dict(autoreset_mode="next_step", time_limit_truncates=True, max_steps=40)),
])
def test_served_steps_are_bit_identical_to_cs_step(task, mode, kw):
    """K served steps (plain rows in through cs_serve_submit, out through cs_serve_collect) against the same
    K steps of cs_step on a twin env: every output of every step and the final state, bit for bit, under
    reset churn, on a ragged batch, for lean and full-featured configurations."""
    import torch
    n, K = 2000 + 37, 240
    served, plain = _twin(task, n, mode, **kw)
    ad = served.action_dim
    g = torch.Generator(device=served.device)
    g.manual_seed(5)
    acts = torch.rand((K, n, ad), generator=g, device=served.device) * 2 - 1
    acts[:, ::3] = HOVER * (1 + 0.02 * torch.randn((K, (n + 2) // 3, ad), generator=g, device=served.device))
    view = served.serve_begin(K, ring=4, timeout=5.0)
    assert (view.tiles, view.obs_dim, view.act_dim, view.num_steps) == ((n + 63) // 64, served.obs_dim, ad, K)
    o0 = served.serve_collect(-1)[0].clone()
    assert torch.equal(o0, plain._obs)                     # the observation before step 0 = what reset returned
    n_done = 0
    for s in range(K):
        served.serve_submit(s, acts[s])
        got = [t.clone() for t in served.serve_collect(s)]
        want = plain.step(acts[s])[:4]
        for k, (x, y) in enumerate(zip(got, want)):
            assert torch.equal(x, y), (s, k)
        n_done += int(got[2].sum()) + int(got[3].sum())
    assert served.serve_end() == K
    assert served.serve_status() == (K, K, 0)
    assert n_done > n // 4 or kw.get("autoreset_mode") == "disabled"      # the churn really happened
    _assert_same_state(served, plain)
    # the env goes on with ordinary steps afterwards
    a = acts[0]
    for x, y in zip(served.step(a)[:4], plain.step(a)[:4]):
        assert torch.equal(x, y)
    served.close()
    plain.close()


def test_served_steps_at_full_size_vs_cs_step_and_oracle():
    """BASELINE config 2's size: 65 536 envs, 1 000 served steps with reset churn, every step bit-identical to
    cs_step on a twin; the first 60 steps also against the CPU oracle."""
    import torch
    n, K = 65536, 1000
    served, orc = make_pair("lander3d", n, "float32", autoreset="next_step", seed=9)
    plain, _ = make_pair("lander3d", n, "float32", autoreset="next_step", seed=9)
    for e in (served, plain, orc):
        e.reset()
    assert served.serve_max_envs() >= n
    g = torch.Generator(device=served.device)
    g.manual_seed(1)
    ring = torch.rand((16, n, 4), generator=g, device=served.device) * 2 - 1
    served.serve_begin(K, ring=8, timeout=5.0)
    bad = torch.zeros((), dtype=torch.int64, device=served.device)
    for s in range(K):
        a = ring[s % 16]
        served.serve_submit(s, a)
        got = served.serve_collect(s)
        want = plain.step(a)[:4]
        for x, y in zip(got, want):
            bad += (x != y).sum()
        if s < 60:
            obs, r, term, trunc = (to_np(t).copy() for t in got)
            assert_step_close((obs, r, term, trunc), orc.step(to_np(a).astype(np.float64)), MODE_TOL["float32"] * 100,
                              r_abs="auto", ctx="step %d" % s)
    assert served.serve_end() == K
    assert int(bad) == 0
    _assert_same_state(served, plain)
    served.close()
    plain.close()


@pytest.mark.parametrize("task,heuristic", [("lander3d", "lander"), ("hover3d", "hover"), ("hover3d", "lander")])
def test_served_closed_loop_policy_kernel_equals_rollout_pid(task, heuristic):
    """A closed loop whose policy is its OWN kernel per step (cs_serve_policy_pid: outputs of step s-1 ->
    PID heuristic -> actions of step s, through the granule rings) against the same loop fused into one
    kernel (cs_rollout_pid): bit-identical outputs, state and controller state."""
    import torch
    n, K = 2500, 160
    a, b = _twin(task, n, "float32", autoreset_mode="next_step")
    for e in (a, b):
        e.configure_pid(heuristic=heuristic)
        e.reset()
    want = [t.clone() for t in b.rollout_pid(K)]
    a.serve_begin(K, ring=2, timeout=5.0)
    outs = [torch.empty_like(t) for t in want]
    for s in range(K):
        a.serve_policy_pid(s)
        a.serve_collect(s, out=(outs[0][s], outs[1][s], outs[2][s].view(torch.uint8), outs[3][s].view(torch.uint8)))
    assert a.serve_end() == K
    for k, (x, y) in enumerate(zip(outs, want)):
        assert torch.equal(x, y), k
    _assert_same_state(a, b)
    # controller state: the fused kernel restarts the controllers of an env that began a new episode in the LAST
    # step before it stores them; the policy kernel does that when it next acts (step K, never launched here)
    pa, pb = a.pid_get_state(), b.pid_get_state()
    reset_last = to_np(outs[2][K - 2] | outs[3][K - 2])           # NEXT_STEP: done at K-2 => reset in step K-1
    assert (pb[:, reset_last] == 0).all()
    pa[:, reset_last] = 0
    assert np.array_equal(pa, pb)
    a.close()
    b.close()


def test_served_persistent_policy_kernel_equals_rollout_pid():
    """The policy of a whole session as ONE kernel next to the env kernel (cs_serve_policy_pid_many: controllers in
    registers, no launch in the loop): 60 closed-loop steps, every output read back from a 64-deep ring afterwards,
    against cs_rollout_pid(60)."""
    import torch
    n, K = 6000, 60
    a, b = _twin("lander3d", n, "float32", autoreset_mode="next_step")
    for e in (a, b):
        e.configure_pid()
        e.reset()
    want = [t.clone() for t in b.rollout_pid(K)]
    a.serve_begin(K, ring=64, timeout=5.0)
    a.serve_policy_pid(0, num_steps=K)
    assert a.serve_end() == K
    outs = [torch.empty_like(t) for t in want]
    for s in range(K):                         # the ring still holds every step of the (closed) session
        a.serve_collect(s, out=(outs[0][s], outs[1][s], outs[2][s].view(torch.uint8), outs[3][s].view(torch.uint8)))
    torch.cuda.current_stream(a.device).synchronize()
    for k, (x, y) in enumerate(zip(outs, want)):
        assert torch.equal(x, y), k
    _assert_same_state(a, b)
    pa, pb = a.pid_get_state(), b.pid_get_state()
    reset_last = to_np(outs[2][K - 2] | outs[3][K - 2])
    pa[:, reset_last] = 0
    assert (pb[:, reset_last] == 0).all() and np.array_equal(pa, pb)
    a.close()
    b.close()


def test_served_session_gives_up_after_its_timeout_and_says_so():
    """A step whose actions never arrive: every wavefront's wait is bounded, the session ends with
    CS_ERR_TIMEOUT, the steps that were served are kept, and the env is usable afterwards."""
    import time
    import torch
    from gym_copter_amd._lib import CopterStepError, ERR_TIMEOUT
    n = 4096
    served, plain = _twin("lander3d", n, "float32", autoreset_mode="next_step")
    acts = torch.full((n, 4), HOVER, dtype=torch.float32, device=served.device)
    served.serve_begin(6, ring=2, timeout=0.25)
    for s in range(3):
        served.serve_submit(s, acts)
        served.serve_collect(s)
        plain.step(acts)
    torch.cuda.current_stream(served.device).synchronize()     # (a DEVICE-wide synchronize would wait for the session)
    t0 = time.perf_counter()
    status = served.serve_status()          # waits for the env kernel: it gives up after 0.25 s
    waited = time.perf_counter() - t0
    assert status == (3, 3, (n + 63) // 64) and 0.2 < waited < 2.0
    served._lib.cs_serve_end(served._ctx, served._stream(), None)
    with pytest.raises(CopterStepError) as ei:
        served.serve_begin(2, timeout=0.05)
        time.sleep(0.3)                     # nothing submitted at all, and nobody stops it in time
        served.serve_end()
    assert ei.value.code == ERR_TIMEOUT
    served.serve_begin(2, timeout=5.0)
    assert served.serve_end() == 0          # closed at once: the stop word ends it, no timeout
    _assert_same_state(served, plain)
    for x, y in zip(served.step(acts)[:4], plain.step(acts)[:4]):
        assert torch.equal(x, y)
    served.close()
    plain.close()


def test_served_session_stops_early_on_request():
    import torch
    n = 3000
    served, plain = _twin("hover3d", n, "float32", autoreset_mode="next_step")
    acts = torch.rand((n, 4), device=served.device)
    served.serve_begin(500, timeout=10.0)
    for s in range(7):
        served.serve_submit(s, acts)
        plain.step(acts)
    assert served.serve_end() == 7          # the stop word ends the session well before the 10 s timeout
    assert served.serve_status() == (7, 7, 0)
    _assert_same_state(served, plain)
    served.close()
    plain.close()


def test_closing_a_context_with_an_open_session_stops_it_first():
    """cs_destroy with a session still open: the env kernel is told to stop and waited for BEFORE its tiles are
    freed (it stores them when it exits), and that costs a poll interval, not the session's 30 s timeout."""
    import time
    import torch
    n = 4096
    import gym_copter_amd
    env = gym_copter_amd.CopterVecEnv(task="lander3d", num_envs=n, seed=2, autoreset_mode="next_step")
    env.reset()
    acts = torch.rand((n, 4), device=env.device)
    env.serve_begin(1000, timeout=30.0)
    for s in range(3):
        env.serve_submit(s, acts)
    torch.cuda.current_stream().synchronize()
    t0 = time.time()
    env.close()
    assert time.time() - t0 < 5.0
    other = gym_copter_amd.CopterVecEnv(task="lander3d", num_envs=n, seed=3)     # the device is fine afterwards
    other.reset()
    other.step(acts)
    torch.cuda.synchronize()
    other.close()


def test_served_feeders_captured_in_a_hipgraph_replay_against_every_session():
    """The K x (submit, collect) launches of one session captured ONCE and replayed against later sessions:
    tags are session-relative and cs_serve_begin zeroes the rings.  begin / end themselves refuse a capturing
    stream (HIP may serialise the branches of one graph: the env kernel must not share one with its feeders)."""
    import torch
    from gym_copter_amd._lib import CopterStepError
    n, K = 8192, 24
    served, plain = _twin("lander3d", n, "float32", autoreset_mode="next_step")
    acts = torch.rand((K, n, 4), device=served.device) * 2 - 1
    outs = (torch.empty((K, n, served.obs_dim), device=served.device), torch.empty((K, n), device=served.device),
            torch.empty((K, n), dtype=torch.uint8, device=served.device),
            torch.empty((K, n), dtype=torch.uint8, device=served.device))

    def feed():
        for s in range(K):
            served.serve_submit(s, acts[s])
            served.serve_collect(s, out=tuple(t[s] for t in outs))

    def expect(tag):
        want = [[t.clone() for t in plain.step(acts[s])[:4]] for s in range(K)]
        for k in range(4):
            got = outs[k].view(torch.bool) if k >= 2 else outs[k]
            assert torch.equal(got, torch.stack([w[k] for w in want])), (tag, k)

    side = torch.cuda.Stream(device=served.device)
    side.wait_stream(torch.cuda.current_stream(served.device))
    with torch.cuda.stream(side):
        served.serve_begin(K, ring=4, timeout=5.0)        # eager session (allocates the rings)
        feed()
        assert served.serve_end() == K
        expect("eager")
        # the feeders of a session of this shape, captured while NO session is open (torch's capture begins
        # with a device-wide synchronize, which an open session's env kernel would sit out until its timeout)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            feed()
            with pytest.raises(CopterStepError):          # sessions are not opened or closed inside a capture
                served.serve_begin(K, ring=4, timeout=5.0)
        for rep in range(3):
            for t in outs:
                t.zero_()
            served.serve_begin(K, ring=4, timeout=5.0)
            graph.replay()
            served.serve_end(wait=False)                  # enqueue only: the next session follows at once
            expect(rep)
            assert served.serve_status() == (K, K, 0)
    torch.cuda.current_stream(served.device).wait_stream(side)
    _assert_same_state(served, plain)
    served.close()
    plain.close()


# ---------------------------------------------------------------------------------------
# a REAL RCCL collective on the one GPU there is: a 1-rank nccl group whose
# all-gathers are issued (force_collective) instead of being shortcut
# ---------------------------------------------------------------------------------------
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_RCCL_CHILD = r"""
import os, sys
sys.path.insert(0, %(root)r)
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
import gym_copter_amd
from gym_copter_amd.sharded import ShardedCopterVecEnv, ShardGather, PackedOutputs
n = 4096
calls = {"n": 0}
real = dist.all_gather_into_tensor
def counted(*a, **k):
    calls["n"] += 1
    return real(*a, **k)
dist.all_gather_into_tensor = counted
g = torch.Generator(device="cuda"); g.manual_seed(1)
acts = torch.rand((12, n, 4), generator=g, device="cuda") * 2 - 1
for gather in ("obs", "all"):
    plain = gym_copter_amd.CopterVecEnv("lander3d", n, seed=5, autoreset_mode="next_step")
    env = ShardedCopterVecEnv("lander3d", total_envs=n, gather=gather, seed=5, autoreset_mode="next_step",
                              force_collective=True)
    assert env.world == 1 and env._gather.force
    env.reset(); plain.reset()
    before = calls["n"]
    for t in range(12):                                   # eager: one RCCL all-gather per step
        for u, v in zip(env.step(acts[t])[:4], plain.step(acts[t])[:4]):
            assert torch.equal(u.reshape(v.shape), v), (gather, t)
    assert calls["n"] - before == 12, calls
    # the same step + collective captured into a hipGraph and replayed
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        env.step(acts[0]); plain.step(acts[0])
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    captured = "graph"
    import time; time.sleep(0.3)        # (the watchdog retires the eager collectives before the capture opens)
    out = None
    try:
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):   # (RCCL's watchdog thread queries events meanwhile)
            out = env.step(acts[1])
    except Exception as e:
        captured = "refused: " + type(e).__name__
        torch.cuda.synchronize()
    if captured == "graph":
        want = [x.clone() for x in plain.step(acts[1])[:4]]
        graph.replay(); torch.cuda.synchronize()
        for u, v in zip(out[:4], want):
            assert torch.equal(u.reshape(v.shape), v), (gather, "replay")
    print("RCCL_LEG", gather, captured)
    env.close(); plain.close()
    # the graph that captured the collective goes BEFORE the communicator does: freed by the garbage collector at
    # interpreter exit, after destroy_process_group, it has ended this process with SIGSEGV (1 run in 10)
    del graph, out, env, plain
    import gc; gc.collect(); torch.cuda.synchronize()
ones = torch.ones(1, device="cuda"); dist.all_reduce(ones)
assert int(ones.item()) == dist.get_world_size() == 1
dist.barrier(); dist.destroy_process_group()
print("RCCL_OK")
"""


def test_a_real_rccl_all_gather_runs_on_one_gpu(tmp_path):
    """force_collective=True: the 1-rank nccl group issues dist.all_gather_into_tensor (counted) for the
    observation rows and for the packed outputs, eagerly and hipGraph-captured, with unchanged results."""
    import subprocess
    import sys
    script = tmp_path / "rccl_child.py"
    script.write_text(_RCCL_CHILD % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29561", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    from gpu_util import run_with_rccl
    p = run_with_rccl([sys.executable, str(script)], env, 240)
    assert p.returncode == 0 and "RCCL_OK" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]
    legs = [ln.split() for ln in p.stdout.splitlines() if ln.startswith("RCCL_LEG")]
    assert [l[1] for l in legs] == ["obs", "all"]
    print("RCCL capture:", legs)


def test_bench_gather_on_one_gpu_reports_what_rccl_saw(tmp_path):
    """`bench.py --gather` on one GPU without a launcher: it opens a 1-rank RCCL group itself, forces the
    collectives, and the JSON line says what RCCL saw and which legs were captured."""
    import json
    import subprocess
    import sys
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5", "--gather",
           "--no-sweep", "--no-cpu-baseline", "--pid", "0", "--many", "0", "--served", "0", "--min-region-ms", "5",
           "--regions", "3", "--full-out", str(tmp_path / "full.json")]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("COPTERSTEP_FORCE_COLLECTIVE", None)
    from gpu_util import run_with_rccl
    p = run_with_rccl(cmd, env, 300, cwd=str(tmp_path))
    assert p.returncode == 0, p.stderr[-4000:]
    from gpu_util import bench_records
    line, d = bench_records(p.stdout, tmp_path / "full.json")
    assert line["rccl"] == d["rccl"] and list(line)[-1] == "summary"
    assert d["rccl"] == {"backend": "nccl", "world_size": 1, "ranks_seen": 1}
    assert d["allgather_is_a_collective"] is True
    assert set(d["allgather_launch_mode"]) == {"obs", "packed", "pipelined"}
    assert d["summary"]["rccl"] == d["rccl"] and list(d)[-1] == "summary"
    assert d["timed_steps_total"] >= 3 * 20 and d["timed_region_s"] > 0
    for k in ("value_with_allgather", "value_with_packed_allgather", "value_with_pipelined_allgather"):
        assert 0 < d[k] <= d["value"] * 1.05, (k, d[k], d["value"])


# ---------------------------------------------------------------------------------------
# north_star's literal bar: per-component PURE-RELATIVE error vs the golden float64 traces
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("mode", ["float32", "float64"])
def test_pure_relative_parity_per_component(mode):
    """|got - ref| / |ref| (masked at |ref| < 1e-3 units) per state component over every step of every golden
    E / V / D / W / R episode of the reference, device env in the default and in the float64 mode.  Asserted:
    the 1e-5 bar holds for all twelve components wherever the reference value is not passing through zero; every
    value above the bar IS such a crossing (parity_report.collect asserts it sample by sample); the float64 mode
    meets the bar everywhere.  The report is printed (pytest -s) and kept under gpurun_out/."""
    import parity_report as pr
    rep = pr.collect(pr.DeviceBackend(mode), float32_inputs_only=True, stride=1 if mode == "float32" else 3)
    text = pr.format_report("device env state_dtype=%s vs golden float64 traces of the reference" % mode, rep)
    print("\n" + text)
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "parity_report_%s.txt" % mode), "w") as f:
            f.write(text + "\n")
    assert rep["samples"] > (200000 if mode == "float32" else 60000)
    if mode == "float64":
        assert rep["worst"].max() <= 1e-9 and rep["over_bar"].sum() == 0
        return
    assert (rep["worst_steady"] <= pr.BAR).all(), rep["worst_steady"]
    assert (rep["worst_range"] <= 1.5e-6).all(), rep["worst_range"]
    over = {pr.NAMES[k]: int(v) for k, v in enumerate(rep["over_bar"]) if v}
    assert set(over) <= {"z", "dx", "dy", "dz", "x", "y"} and "z" in over, over
    assert rep["worst"][4] < 1e-3 and (rep["worst"][6:] <= pr.BAR).all()


def test_served_session_fed_from_many_streams():
    """HIP multiplexes streams onto a few hardware queues; a feeder stream that shared the env kernel's queue would
    sit behind the persistent kernel and dead-lock the session.  The env kernel's stream is the only
    high-priority stream: sessions fed from 24 different default-priority streams all complete."""
    import torch
    n, K = 4096, 12
    served, plain = _twin("lander3d", n, "float32", autoreset_mode="next_step")
    acts = torch.rand((K, n, 4), device=served.device) * 2 - 1
    streams = [torch.cuda.Stream(device=served.device) for _ in range(24)]
    for st in streams:
        st.wait_stream(torch.cuda.current_stream(served.device))
        with torch.cuda.stream(st):
            served.serve_begin(K, ring=2, timeout=0.5)
            for s in range(K):
                served.serve_submit(s, acts[s])
                served.serve_collect(s)
            assert served.serve_end() == K
        torch.cuda.current_stream(served.device).wait_stream(st)
        for s in range(K):
            plain.step(acts[s])
    _assert_same_state(served, plain)
    served.close()
    plain.close()


def test_served_session_argument_errors():
    """What cs_serve_* refuses, as error codes with messages: a batch whose wavefronts could not all stay resident,
    a ring that is not a power of two, a second session, feeders before any session, the PID policy on a task it
    does not fly."""
    import torch
    import gym_copter_amd
    from gym_copter_amd._lib import CopterStepError
    big = gym_copter_amd.CopterVecEnv("lander3d", 1 << 20)
    assert big.serve_max_envs() < (1 << 20)
    with pytest.raises(CopterStepError, match="resident"):
        big.serve_begin(4)
    big.close()
    env = gym_copter_amd.CopterVecEnv("lander1d", 4096)
    env.reset()
    a = torch.zeros((4096, 1), device=env.device)
    with pytest.raises(CopterStepError, match="no session"):
        env.serve_submit(0, a)
    with pytest.raises(CopterStepError, match="power of two"):
        env.serve_begin(4, ring=3)
    env.serve_begin(4, ring=2, timeout=1.0)
    with pytest.raises(CopterStepError, match="already open"):
        env.serve_begin(4)
    with pytest.raises(CopterStepError, match="step must be"):
        env.serve_submit(4, a)
    for call in (lambda: env.step(a), env.reset, env.get_state, lambda: env.rollout_random(2)):
        with pytest.raises(CopterStepError, match="served session is open"):     # the state is in the kernel's registers
            call()
    for s in range(4):
        env.serve_submit(s, a)
    assert env.serve_end() == 4
    with pytest.raises(CopterStepError, match="3D"):      # (after the session: configuring allocates and synchronises)
        env.serve_policy_pid(0)
    env.close()


def test_served_session_under_uneven_load():
    """The hand-offs under load: while another stream streams 2 GB through the memory system again and again (every
    CU busy with loads and stores, the L2s churning), a served session at 65 536 envs still delivers every word of
    every step bit-identically to cs_step -- a stale or torn granule would show up as a different number."""
    import torch
    n, K = 65536, 300
    served, plain = _twin("hover3d", n, "float32", autoreset_mode="next_step")
    g = torch.Generator(device=served.device)
    g.manual_seed(3)
    acts = torch.rand((8, n, 4), generator=g, device=served.device) * 2 - 1
    big = torch.empty(1 << 28, dtype=torch.float32, device=served.device)      # 1 GiB
    other = torch.empty_like(big)
    noise = torch.cuda.Stream(device=served.device)
    feed = torch.cuda.Stream(device=served.device)
    stop_at = 40
    with torch.cuda.stream(noise):
        for _ in range(stop_at):
            other.copy_(big)                  # ~0.35 ms each at ~6 TB/s: keeps the memory system saturated
            big.add_(1.0)
    bad = torch.zeros((), dtype=torch.int64, device=served.device)
    feed.wait_stream(torch.cuda.current_stream(served.device))
    with torch.cuda.stream(feed):
        served.serve_begin(K, ring=4, timeout=10.0)
        for s in range(K):
            a = acts[s % 8]
            served.serve_submit(s, a)
            got = served.serve_collect(s)
            want = plain.step(a)[:4]
            for x, y in zip(got, want):
                bad += (x != y).sum()
        assert served.serve_end() == K
    noise.synchronize()
    torch.cuda.current_stream(served.device).wait_stream(feed)
    assert int(bad) == 0
    _assert_same_state(served, plain)
    served.close()
    plain.close()


def test_float64_mode_keeps_float64_forces():
    """with float64 state words an installed force (reset options / Dynamics.perturb) stays
    float64 -- upstream's force / M is -- instead of passing through the float32 device rows."""
    n = 200
    env, orc = make_pair("lander3d", n, "float64")
    rng = np.random.default_rng(8)
    f = rng.uniform(-30, 30, (3, n)) + 1.0 / 3.0                      # not float32-representable
    env.reset(options={"forces": f})
    orc.reset(forces=f)
    assert np.array_equal(env.get_state()["force"], f)
    got, want, _ = step_both(env, orc, np.full((n, 4), HOVER, np.float32))
    assert_step_close(got, want, MODE_TOL["float64"])
    assert_state_close(env, orc, MODE_TOL["float64"])
    f2 = rng.uniform(-5, 5, (3, n)) + 1.0 / 7.0
    env.set_perturbation(f2)
    assert np.array_equal(env.get_state()["force"], f2) and (env.get_state()["flags"] & 5 == 5).all()
    m = np.arange(n) % 2 == 0
    env.reset(options={"forces": f, "mask": m})
    st = env.get_state()
    assert np.array_equal(st["force"][:, m], f[:, m]) and np.array_equal(st["force"][:, ~m], f2[:, ~m])
    env.close()


def test_served_long_soak_closed_loop_and_graph_fed():
    """Integrity of the hand-offs over many transfers: (1) 50 000 closed-loop steps at 65 536 envs with the policy
    and the env both persistent (4e11 granule words through device memory) end in exactly the state, controller state
    and episode counters that cs_rollout_pid reaches -- one stale or torn granule anywhere would change them;
    (2) 20 graph-fed sessions of 500 steps, every output word compared on the device against cs_step."""
    import torch
    n, K = 65536, 50000
    a, b = _twin("lander3d", n, "float32", autoreset_mode="next_step")
    for e in (a, b):
        e.configure_pid()
        e.reset()
    for _ in range(K // 1000):
        b._lib.cs_rollout_pid(b._ctx, 1000, None, None, None, None, None, b._stream())
    a.serve_begin(K, ring=4, timeout=10.0)
    a.serve_policy_pid(0, num_steps=K)
    assert a.serve_end() == K
    _assert_same_state(a, b)
    pa, pb = a.pid_get_state(), b.pid_get_state()
    same = (pa == pb).all(axis=0)
    assert same.mean() > 0.98 and (pb[:, ~same] == 0).all()      # (envs reset in the very last step: see the K-step test)
    # (2) plain rows through submit / collect, feeders replayed from one hipGraph
    K2 = 500
    acts = torch.rand((K2, n, 4), device=a.device) * 2 - 1
    outs = (torch.empty((K2, n, a.obs_dim), device=a.device), torch.empty((K2, n), device=a.device),
            torch.empty((K2, n), dtype=torch.uint8, device=a.device), torch.empty((K2, n), dtype=torch.uint8, device=a.device))
    ref = tuple(torch.empty_like(t) for t in outs)
    a.serve_begin(K2, ring=4, timeout=10.0)               # (allocation for this shape, outside the capture)
    a.serve_end()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, capture_error_mode="thread_local"):
        for s in range(K2):
            a.serve_submit(s, acts[s])
            a.serve_collect(s, out=tuple(t[s] for t in outs))
    bad = torch.zeros((), dtype=torch.int64, device=a.device)
    for rep in range(20):
        a.serve_begin(K2, ring=4, timeout=10.0)
        graph.replay()
        a.serve_end(wait=False)
        for s in range(K2):
            b.bind_outputs(ref[0][s], ref[1][s], ref[2][s], ref[3][s])
            b.step(acts[s])
        for x, y in zip(outs, ref):
            bad += (x != y).sum()
    assert a.serve_status() == (K2, K2, 0) and int(bad) == 0
    _assert_same_state(a, b)
    a.close()
    b.close()


def test_caller_side_policy_kernel_on_the_public_device_header():
    """tests/host/serve_policy_host.hip: a third party's HIP policy kernel built only on include/copterstep.h +
    include/copterstep_serve.h, one launch per closed-loop step against a served session, checked against a twin
    stepped with cs_step on the actions the policy recorded."""
    import subprocess
    exe = os.path.join(ROOT, "tests", "host", "serve_policy_host")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    p = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, (p.returncode, p.stdout, p.stderr)
    assert "serve_policy_host: OK" in p.stdout


def test_caller_side_policy_fused_into_the_k_step_kernel():
    """tests/host/rollout_policy_host.hip: the caller's OWN policy as a device functor, instantiated into the K-step
    kernel in the caller's translation unit (include/copterstep_rollout.h on cs_get_launch_view): a replay policy is
    bit-identical to cs_step_many, a closed-loop policy with per-env state is bit-identical to a twin stepped with
    cs_step on the recorded actions (and the host re-evaluates the law from what the twin returned), on the lean and
    on the full-featured instantiation; the wrong task is refused.  Prints us per env step at 65 536 envs."""
    import subprocess
    exe = os.path.join(ROOT, "tests", "host", "rollout_policy_host")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    p = subprocess.run([exe, "time"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, (p.returncode, p.stdout, p.stderr)
    assert "rollout_policy_host: OK" in p.stdout
    print(p.stdout)


def test_launch_view_describes_the_context():
    import ctypes as C
    import gym_copter_amd
    from gym_copter_amd import _lib
    for kw, lean in ((dict(), 1), (dict(episode_stats=True), 0), (dict(substeps=3), 1)):
        env = gym_copter_amd.CopterVecEnv(task="hover3d", num_envs=1000, state_dtype="float64", **kw)
        v = _lib.LaunchView()
        # struct_size is an in-parameter: a caller built against another layout is refused and not written to
        v.struct_size, v.grid = C.sizeof(_lib.LaunchView) - 8, 777
        assert env._lib.cs_get_launch_view(env._ctx, C.byref(v)) == _lib.ERR_ABI and v.grid == 777
        assert "struct_size" in env._lib.cs_last_error().decode()
        v.struct_size = C.sizeof(_lib.LaunchView)
        _lib.check(env._lib.cs_get_launch_view(env._ctx, C.byref(v)))
        assert (v.struct_size, v.abi_version) == (C.sizeof(_lib.LaunchView), _lib.ABI_VERSION)
        assert (v.task, v.state_mode, v.num_envs, v.grid, v.block) == (_lib.TASK_HOVER3D, _lib.STATE_F64, 1000, 16, 64)
        assert v.lean == lean and v.one_call == (0 if "substeps" in kw else 1) and v.direct_rows == 1
        assert v.consts and v.state and v.consts_size > 256 and v.state_size >= 32
        env.serve_begin(2, timeout=1.0)                      # refused while a served session is open
        assert env._lib.cs_get_launch_view(env._ctx, C.byref(v)) == _lib.ERR_ARG
        env._lib.cs_set_last_error(b"said by a caller-side header")     # (what copterstep_rollout.h's refusals use)
        assert env._lib.cs_last_error() == b"said by a caller-side header"
        env.serve_end(wait=False)
        env.close()


def test_numpy_returns_are_the_callers_to_keep_unless_copy_is_off():
    """gymnasium.vector.SyncVectorEnv(copy=True) semantics on the NumPy convenience path: by default what step()
    returned is not touched by later steps; copy=False hands out views of two alternating pinned buffers."""
    import gym_copter_amd
    n = 1000
    rng = np.random.default_rng(5)
    acts = [rng.uniform(-1, 1, (n, 4)).astype(np.float32) for _ in range(4)]
    for copy in (True, False):
        env = gym_copter_amd.CopterVecEnv(task="lander3d", num_envs=n, seed=9, copy=copy)
        twin = gym_copter_amd.CopterVecEnv(task="lander3d", num_envs=n, seed=9)
        env.reset()
        twin.reset()
        kept, want = [], []
        for a in acts:
            kept.append(env.step(a)[:4])
            want.append(tuple(np.array(v) for v in twin.step(a)[:4]))
        assert all(isinstance(v, np.ndarray) for v in kept[0])
        for j in range(len(acts)):
            same = all(np.array_equal(k, w) for k, w in zip(kept[j], want[j]))
            if copy or j >= len(acts) - 2:
                assert same, (copy, j)             # copy=False: the last two steps' views are still intact
        if not copy:
            assert np.shares_memory(kept[0][0], kept[2][0]) and not np.shares_memory(kept[0][0], kept[1][0])
        env.close()
        twin.close()


_LINEAR_POLICY = """
struct Policy {
  const float* params;                 // [ACT][OBS] weights, then ACT biases, then one thrust trim per env
  float w[ACT * OBS + ACT];
  float trim;
  __device__ void load(uint32_t env, bool valid) {
    for (int j = 0; j < ACT * OBS + ACT; ++j) w[j] = params[j];
    trim = valid ? params[ACT * OBS + ACT + env] : 0.f;
  }
  __device__ void store(uint32_t, bool) {}
  __device__ void operator()(const float (&obs)[OBS], uint32_t, int, bool, float (&a)[ACT]) const {
    for (int m = 0; m < ACT; ++m) {
      float s = w[ACT * OBS + m];
      for (int j = 0; j < OBS; ++j) s += w[m * OBS + j] * obs[j];
      a[m] = s + trim;
    }
  }
};
"""


@pytest.mark.parametrize("task,mode", [("lander3d", "float32"), ("hover2d", "float64")])
def test_python_callers_policy_source_is_compiled_and_fused(task, mode, tmp_path):
    """gym_copter_amd.compile_policy + env.rollout_policy: a policy given as HIP source (a linear law with shared
    weights and one parameter per env) is compiled with hipcc at run time, fused into the K-step kernel and flown
    closed-loop for K steps in one launch.  Checked against a twin env stepped with step() on the actions the policy
    recorded (bit-identical outputs and state: the loop is closed and the fused kernel IS the step), and the law
    itself against NumPy on the observations returned."""
    import shutil
    import torch
    import gym_copter_amd
    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("compile_policy needs hipcc on the box")
    n, K = 1500, 40
    mk = lambda: gym_copter_amd.CopterVecEnv(task=task, num_envs=n, state_dtype=mode, seed=5, autoreset_mode="next_step",
                                             max_steps=25)
    env, twin = mk(), mk()
    od, ad = env.obs_dim, env.action_dim
    rng = np.random.default_rng(3)
    W = (rng.standard_normal((ad, od)) * 1e-3).astype(np.float32)
    b = np.full(ad, HOVER, np.float32)
    trim = (rng.standard_normal(n) * 2e-4).astype(np.float32)
    params = torch.from_numpy(np.concatenate([W.ravel(), b, trim])).to(env.device)
    policy = gym_copter_amd.compile_policy(env, _LINEAR_POLICY, cache_dir=str(tmp_path))
    again = gym_copter_amd.compile_policy(env, _LINEAR_POLICY, cache_dir=str(tmp_path))     # served from the cache
    assert again.path == policy.path and os.path.exists(policy.path)
    obs0, _ = env.reset()
    obs0 = to_np(obs0).copy()
    twin.reset()
    obs, rew, term, trunc, acts = (to_np(v).copy() for v in env.rollout_policy(policy, K, params, return_actions=True))
    seen = obs0
    for k in range(K):
        want = (seen.astype(np.float64) @ W.T.astype(np.float64) + b + trim[:, None]).astype(np.float32)
        assert np.allclose(acts[k], want, rtol=1e-5, atol=1e-7), k
        o, r, t, u, _ = twin.step(torch.from_numpy(acts[k]).to(twin.device))
        assert np.array_equal(to_np(o), obs[k]) and np.array_equal(to_np(r), rew[k]), k
        assert np.array_equal(to_np(t), term[k]) and np.array_equal(to_np(u), trunc[k]), k
        seen = obs[k]
    assert term.any() or trunc.any()                     # episodes ended and restarted inside the launch
    _assert_same_state(env, twin)
    other = gym_copter_amd.CopterVecEnv(task="lander1d", num_envs=64)
    with pytest.raises(ValueError):
        other.rollout_policy(policy, 2, params)
    with pytest.raises(RuntimeError, match="hipcc failed"):
        gym_copter_amd.compile_policy(env, "struct Policy { this is not HIP };", cache_dir=str(tmp_path))
    for e in (env, twin, other):
        e.close()
