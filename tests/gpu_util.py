"""Helpers shared by the -m gpu parity tests: build a device env and the CPU checker
(oracle.refvec.VecOracle) with the same configuration, and compare them."""
import numpy as np

from oracle import refvec
from oracle.refcpu import TaskParams
from oracle.refvec import VecOracle

AUTORESET = {"disabled": refvec.AUTORESET_DISABLED, "next_step": refvec.AUTORESET_NEXT_STEP,
             "same_step": refvec.AUTORESET_SAME_STEP}

# per-component scale of the parity metric |got - ref| <= tol * max(|ref|, scale):
# 1 m, 1 m/s, 1 rad, 1 rad/s (a pure relative test is ill-posed at zero crossings)
SCALE = 1.0


def have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def make_pair(task="lander3d", n=1, mode="float32", autoreset="disabled", substeps=1, seed=0,
              env_id_base=0, time_limit_truncates=False, episode_stats=False, **task_kwargs):
    import gym_copter_amd
    env = gym_copter_amd.CopterVecEnv(task=task, num_envs=n, state_dtype=mode,
                                      autoreset_mode=autoreset, substeps=substeps, seed=seed,
                                      env_id_base=env_id_base,
                                      time_limit_truncates=time_limit_truncates,
                                      episode_stats=episode_stats, **task_kwargs)
    # (model switches of the device env that the task-parameter record of the oracle does not hold)
    tp = TaskParams(**{k: v for k, v in task_kwargs.items()
                       if k not in ("action_arith", "thrust_model", "rotor_gyro", "vehicle_params", "world_params",
                                    "track_time")})
    orc = VecOracle(task, n, tp, substeps=substeps, store_mode=mode, autoreset=AUTORESET[autoreset],
                    seed=seed, env_id_base=env_id_base, time_limit_truncates=time_limit_truncates)
    return env, orc


def scaled_err(got, ref):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    with np.errstate(invalid="ignore"):
        e = np.abs(got - ref) / np.maximum(np.abs(ref), SCALE)
    both_nan = np.isnan(got) & np.isnan(ref)
    same_inf = np.isinf(got) & np.isinf(ref) & (np.sign(got) == np.sign(ref))
    e = np.where(both_nan | same_inf, 0.0, e)
    return float(np.nanmax(e)) if e.size else 0.0


def to_np(t):
    return t.detach().cpu().numpy() if hasattr(t, "detach") else np.asarray(t)


def step_both(env, orc, actions):
    """actions: numpy float32 [n,4].  Returns ((obs,r,term,trunc) device-as-numpy, oracle tuple)."""
    import torch
    a = torch.from_numpy(np.ascontiguousarray(actions, dtype=np.float32)).to(env.device)
    obs, r, term, trunc, infos = env.step(a)
    got = tuple(to_np(v).copy() for v in (obs, r, term, trunc))
    want = orc.step(actions.astype(np.float64))
    return got, want, infos


def reward_limit(wobs, wr, r_abs=2e-3, r_rel=2e-6):
    """Per-env tolerance of the float32 reward against the oracle's float64 one: the golden-trace formula
    5e-5 + 1e-5 |r| + 6e-7 |prev_shaping| (tests/test_gpu_golden.py: prev_shaping is a float32 word, so the reward
    carries its rounding), with |prev_shaping| bounded from what the step returned: |shaping now| + |r|, shaping now
    <= 25 * |first six observed components| + 110 (the |dz| penalty and a yaw term the observation does not show).
    About 2.7e-4 for an env at altitude 10; never looser than the (r_abs, r_rel) a test passes -- that pair is what
    full-throttle lanes with shaping in the thousands get, selected by their magnitude, not globally."""
    wobs = np.asarray(wobs, dtype=np.float64)
    wr = np.abs(np.asarray(wr, dtype=np.float64))
    with np.errstate(invalid="ignore", over="ignore"):
        s = wr + 25.0 * np.sqrt(np.sum(np.square(wobs[:, :6]), axis=1)) + 110.0
    s = np.where(np.isfinite(s), s, np.inf)
    return np.minimum(5e-5 + 1e-5 * wr + 6e-7 * s, r_abs + r_rel * wr)


def assert_step_close(got, want, x_tol, r_abs=5e-5, r_rel=1e-5, ctx="", r_unit=0.0, shaping=None):
    """r_abs = "auto": the magnitude-aware reward tolerance of reward_limit() (bounded by 2e-3 + 2e-6 |r|).
    r_unit, shaping: with a numeric r_abs, an extra r_unit x (|shaping| + |r|), `shaping` being the ORACLE's shaping
    value per env (the larger of before / after the step) -- the reward is a difference of two shaping values computed
    from the state (one of them a stored word): its error scales with |shaping|, not with |r|."""
    obs, r, term, trunc = got
    wobs, wr, wterm, wtrunc = want
    assert np.array_equal(term.astype(bool), wterm), "terminated mismatch %s" % ctx
    assert np.array_equal(trunc.astype(bool), wtrunc), "truncated mismatch %s" % ctx
    e = scaled_err(obs, wobs)
    assert e <= x_tol, "obs err %.3e > %.1e %s" % (e, x_tol, ctx)
    dr = np.abs(r.astype(np.float64) - wr)
    lim = reward_limit(wobs, wr) if isinstance(r_abs, str) else r_abs + r_rel * np.abs(wr)
    if r_unit and shaping is not None and not isinstance(r_abs, str):
        s = np.abs(wr) + np.abs(np.nan_to_num(np.asarray(shaping, dtype=np.float64), nan=0.0, posinf=np.inf, neginf=np.inf))
        lim = lim + r_unit * s
    j = int(np.argmax(dr - lim))
    assert np.all(dr <= lim), "reward err %.3e at env %d (got %r want %r term %r obs %r) %s" % (
        float(dr[j]), j, r[j], wr[j], term[j], obs[j], ctx)


def assert_state_close(env, orc, x_tol, ctx=""):
    s = env.get_state()
    assert np.array_equal(s["status"], orc.status), "status mismatch %s" % ctx
    assert np.array_equal(s["steps"], orc.steps), "steps mismatch %s" % ctx
    assert np.array_equal((s["flags"] & 1).astype(bool), orc.pending), "pending flag %s" % ctx
    e = scaled_err(s["x"], orc.x)
    assert e <= x_tol, "state err %.3e > %.1e %s" % (e, x_tol, ctx)
    return e


# tolerance of "device vs the oracle run in the SAME storage mode" (only float64 rounding
# differences: fma contraction, reciprocal-multiply, sin/cos ulps) per mode
MODE_TOL = {"float64": 1e-11, "float32": 2e-8, "float32_rn": 5e-6}


def run_with_rccl(cmd, env, timeout, cwd=None):
    """subprocess.run for a child that opens an RCCL process group.  One retry for TWO known third-party failures,
    both in ProcessGroupNCCL's watchdog thread (PyTorch's code, not this repository's), seen about once in fifteen
    cold starts on the GPU box: the watchdog aborting the process (SIGABRT, 'ProcessGroupNCCL' + 'Watchdog' in the
    child's stderr) when its event query races a stream capture or the teardown, and -- rarer -- the child not
    coming back at all.  A child that overruns `timeout` is sent SIGABRT (PYTHONFAULTHANDLER=1: its Python stacks
    land in stderr and are printed here), then retried once.  Anything else fails at once."""
    import signal
    import subprocess
    env = dict(env, PYTHONFAULTHANDLER="1")
    # a world of one on this host: keep RCCL's bootstrap off whatever other interfaces the box has (probing an
    # unreachable one costs minutes), as tests/test_gpu_multigpu.py::test_c_host_rccl_allgather does for the C host
    env.setdefault("NCCL_SOCKET_IFNAME", "lo")
    env.setdefault("NCCL_IB_DISABLE", "1")

    def once():
        child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=cwd)
        try:
            out, err = child.communicate(timeout=timeout)
            return subprocess.CompletedProcess(cmd, child.returncode, out, err), False
        except subprocess.TimeoutExpired:
            child.send_signal(signal.SIGABRT)
            try:
                out, err = child.communicate(timeout=30)
            except subprocess.TimeoutExpired:
                child.kill()
                out, err = child.communicate()
            return subprocess.CompletedProcess(cmd, -signal.SIGABRT, out, "TIMEOUT after %d s\n%s" % (timeout, err)), True

    p, hung = once()
    if hung:
        print("note: RCCL child did not return in %d s; its stacks:\n%s\nretrying once" % (timeout, p.stderr[-6000:]))
        p, _ = once()
    elif p.returncode != 0 and "ProcessGroupNCCL" in p.stderr and "Watchdog" in p.stderr:
        print("note: ProcessGroupNCCL watchdog abort in the child (rc %d); retrying once" % p.returncode)
        p, _ = once()
    return p


def device_steps_cap(max_steps):
    """The device's step counter saturates at 2^S - 1, S = bits of 2 * (max_steps + 1) (copterstep_internal.h:
    steps_bits_for; include/copterstep.h, cs_config.max_steps) -- 2047 at the default limit of 1000.  Upstream's never
    does (task.py:130) and neither does the oracle's: a test that runs an env nobody resets past the cap compares
    np.minimum(oracle.steps, device_steps_cap(max_steps))."""
    s = 1
    while (1 << s) - 1 < 2 * (max_steps + 1):
        s += 1
    return (1 << s) - 1


def device_episode_bits(max_steps):
    """Bits of the episode counter that live in the device's meta word (29 - S); the rest is in the tile's EPH row."""
    s = 1
    while (1 << s) - 1 < 2 * (max_steps + 1):
        s += 1
    return 29 - s


def bench_records(stdout, full_path):
    """bench.py's two outputs -> (compact line as a dict, full record as a dict).  Asserts what a bounded reader of
    stdout relies on: ONE line, at most 8 000 bytes, nothing after it, and that the line agrees with the full record
    on the contract values."""
    import json
    lines = [ln for ln in stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and stdout.endswith("\n"), stdout[-2000:]
    assert len(lines[0]) <= 8000, len(lines[0])
    line = json.loads(lines[0])
    full = json.load(open(full_path))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "scaling", "dtype", "status"):
        assert line[k] == full[k], k
    assert line["roofline"]["frac"] == full["roofline"]["frac"]
    return line, full
