"""-m gpu: served stepping (cs_serve_*): a persistent env kernel fed through device memory -- bit-identical to cs_step,
closed loop against cs_rollout_pid, time-outs, early stops, draining, hipGraph-replayed feeders, the stop-word race."""
import os
import subprocess

import numpy as np
import pytest

from gpu_util import (MODE_TOL, assert_step_close, have_gpu, make_pair, to_np)

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not have_gpu(), reason="needs a HIP device")]

HOVER = float(np.load(os.path.join(os.path.dirname(__file__), "golden", "meta.npz"))["hover_motor"])
# ---------------------------------------------------------------------------------------
# a REAL RCCL collective on the one GPU there is: a 1-rank nccl group whose
# all-gathers are issued (force_collective) instead of being shortcut
# ---------------------------------------------------------------------------------------
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _assert_same_state(a, b):
    sa, sb = a.get_state(), b.get_state()
    for k in sa:
        assert np.array_equal(sa[k], sb[k], equal_nan=True), k


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


def test_served_collect_writes_interleaved_flags():
    """cs_serve_collect into the wrapper's default (interleaved) flag buffers against cs_step on a twin."""
    import torch
    import gym_copter_amd
    n, K = 700, 10
    kw = dict(task="lander3d", num_envs=n, seed=5, autoreset_mode="next_step", max_steps=6)
    a_env, b_env = gym_copter_amd.CopterVecEnv(**kw), gym_copter_amd.CopterVecEnv(**kw)
    a_env.reset()
    b_env.reset()
    g = torch.Generator(device=a_env.device)
    g.manual_seed(1)
    acts = torch.rand((K, n, 4), generator=g, device=a_env.device) * 0.04
    a_env.serve_begin(K, timeout=5.0)
    fired = 0
    for s in range(K):
        a_env.serve_submit(s, acts[s])
        got = a_env.serve_collect(s)
        want = b_env.step(acts[s])
        for k in range(4):
            assert torch.equal(got[k], want[k]), (s, k)
        fired += int(got[2].sum())
    assert a_env.serve_end() == K and fired > 0
    a_env.close()
    b_env.close()


# ---------------------------------------------------------------------------------------
# a closed-but-running session, feeders without a session, signed zeros across call forms
# ---------------------------------------------------------------------------------------
def test_a_session_closed_without_waiting_is_drained_before_other_streams_touch_the_tiles():
    """cs_serve_end(wait=False) orders only ITS stream behind the env kernel's exit; a step enqueued right away on
    ANOTHER stream must still see the tiles the session wrote back.  The context stays 'draining' until the exit has
    been observed, and every entry point orders its own stream behind it."""
    import torch
    import gym_copter_amd
    n, K = 65536, 40
    kw = dict(task="lander3d", num_envs=n, seed=8, autoreset_mode="next_step")
    env, twin = gym_copter_amd.CopterVecEnv(**kw), gym_copter_amd.CopterVecEnv(**kw)
    env.reset()
    twin.reset()
    g = torch.Generator(device=env.device)
    g.manual_seed(4)
    acts = torch.rand((K + 1, n, 4), generator=g, device=env.device) * 2 - 1
    other = torch.cuda.Stream(device=env.device)
    torch.cuda.synchronize()
    for rep in range(3):
        env.serve_begin(K, ring=8, timeout=5.0)
        for s in range(K):
            env.serve_submit(s, acts[s])
        env.serve_end(wait=False)                  # the env kernel is still working through its ring
        # `other` is NOT ordered behind the current stream (the action tensors have long been written): the only
        # thing that keeps its step behind the env kernel's write-back is the context's draining state
        with torch.cuda.stream(other):
            got = [t.clone() for t in env.step(acts[K])[:4]]
        for s in range(K):
            twin.step(acts[s])
        want = twin.step(acts[K])[:4]
        other.synchronize()
        for k in range(4):
            if not torch.equal(got[k], want[k]):       # say WHAT differs: whole tiles (a stale tile) or single rows
                d = (got[k] != want[k]).reshape(n, -1).any(dim=1)
                idx = torch.nonzero(d).flatten().cpu().numpy()
                raise AssertionError("rep %d output %d: %d envs differ, tiles %s, serve_status %r" % (
                    rep, k, idx.size, np.unique(idx // 64)[:16], env.serve_status()))
    assert env.serve_status() == (K, K, 0)
    sa, sb = env.get_state(), twin.get_state()
    for k in sa:
        assert np.array_equal(sa[k], sb[k], equal_nan=True), k
    # feeders launched eagerly with no session open are refused instead of polling until their timeout
    from gym_copter_amd import _lib
    env.serve_collect(K - 1)                       # (reading a closed session's output ring stays allowed)
    for call in (lambda: env.serve_submit(0, acts[0]), lambda: env.serve_policy_pid(0)):
        with pytest.raises(_lib.CopterStepError, match="no session is open|cs_pid_configure"):
            call()
    env.close()
    twin.close()


@pytest.mark.parametrize("n,K,ring,sessions", [(65536, 1, 2, 20000), (4096, 1, 2, 500), (65536, 3, 2, 1000), (65536, 12, 4, 150)])
def test_a_session_stopped_right_behind_its_last_row_still_takes_that_row(n, K, ring, sessions):
    """cs_serve_end raises the stop word BEHIND everything the caller enqueued: a row submitted before it must be
    stepped, however closely the stop word follows it.  The env kernel used to look at the stop word after a (possibly
    stale) look at the row and gave up on a row that had landed in between -- seen once as a mismatch in
    test_a_session_closed_without_waiting_is_drained_before_other_streams_touch_the_tiles; now the row as it reads AFTER
    the stop word was seen decides (copterstep_serve.hip).  Many short sessions, all rows submitted at once, closed
    without waiting: every tile completes every step, and the envs end where a plain twin ends.  (The old wait loop
    lost a tile's step in 5 of 20 000 one-step sessions at 65 536 envs, none at smaller batches:
    profiles/r05_serve_stop_race.txt -- hence 20 000 sessions of that shape, under three seconds.)"""
    import torch
    import gym_copter_amd
    kw = dict(task="lander3d", num_envs=n, seed=8, autoreset_mode="next_step")
    env, twin = gym_copter_amd.CopterVecEnv(**kw), gym_copter_amd.CopterVecEnv(**kw)
    env.reset()
    twin.reset()
    g = torch.Generator(device=env.device)
    g.manual_seed(4)
    acts = torch.rand((K, n, 4), generator=g, device=env.device) * 2 - 1
    for s in range(sessions):
        env.serve_begin(K, ring=ring, timeout=5.0)
        for k in range(K):
            env.serve_submit(k, acts[k])
        env.serve_end(wait=False)
        assert env.serve_status() == (K, K, 0), s
    for s in range(sessions):
        for k in range(K):
            twin.step(acts[k])
    sa, sb = env.get_state(), twin.get_state()
    for k in sa:
        assert np.array_equal(sa[k], sb[k], equal_nan=True), k
    env.close()
    twin.close()
