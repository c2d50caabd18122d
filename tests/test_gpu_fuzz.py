"""-m gpu: seeded combinatorial parity fuzz.  The other GPU files sweep one option at a time; here every case draws a
whole configuration at random -- task x storage mode x auto-reset mode x substeps x frame rate x time-limit handling
x episode statistics x tick counter x batch shape x shard offset x task constants -- and flies it for 90 steps
against the CPU oracle (oracle/refvec.py, itself pinned bit-for-bit to the reference's golden traces), changing the
STEPPING FORM every few steps: one launch per step (cs_step), K steps per launch (cs_step_many), a served session
(cs_serve_*) and, in every fifth case (all six tasks), a caller-side replay policy fused into the K-step kernel (compile_policy +
rollout_policy: the kernel of include/copterstep_rollout.h instantiated for that case's task and storage mode).  Every step's observation, reward and flags, and the stored state, status, counters (and ticks) after
every stretch are compared; discrete outputs exactly, the state to the same-mode tolerance (MODE_TOL), the float32
rows to one ulp.

Seeds 0..63 run in the suite (device and oracle are deterministic: the cases are fixed).  A one-off sweep of seeds
64..3199 (tools/fuzz_sweep.py; round 5, profiles/r05_fuzz_sweep.txt) passes 3 122 of 3 136 -- every float64 and
float32_rn case, every case with an auto-reset mode, every case of the gentle action laws.  The fourteen others (181 and
360 known since round 3) are not bugs but chaos, all of ONE class: default float32 words, auto-reset off, full-range
actions -- a finished env tumbles on at hundreds of m/s until the stretch ends, and one unit of the stored format grows
to 2.0-8.4e-8 on ONE lane within the stretch (tools/fuzz_trace.py shows it step by step).  Where that unit comes from is
pinned down in tests/test_gpu_numerics.py: NOT the stored format (the codec is bit-exact against the oracle's model over
2.1 M values) but the float64 value that is rounded into it -- the float32 modes' shorter sin / cos polynomials move dx,
dy, dz by ~1e-11, which straddles a rounding boundary of the 29-bit format about once in 1e4 values (measured 9.9e-5).
Two more seeds of that sweep (2102, 2256) showed the same unit through the REWARD (REWARD_UNIT below) and run in the
suite since.

What the reference offers for this: nothing (it has no tests); the oracle is the reference's algorithm
(envs/task.py:77-137, dynamics/__init__.py:114-197, envs/lander.py:46-74)."""
import numpy as np
import pytest

from gpu_util import MODE_TOL, assert_state_close, assert_step_close, have_gpu, make_pair, to_np

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not have_gpu(), reason="needs a HIP device")]

TASKS = ["lander3d", "hover3d", "lander2d", "lander1d", "hover2d", "hover1d"]
OBS_ULP = 1.2e-7       # one float32 ulp relative to max(|ref|, 1)
# reward = shaping - prev_shaping, and prev_shaping is kept as a float32 word in the float32 modes (a float64 in the
# float64 mode), device and oracle alike: where the two states differ by one unit of the stored format the two
# float64 shaping values differ a little, which now and then (2.5 % of such lanes at 120 m/s) rounds the WORD the other
# way -- one float32 ulp of |shaping|, 2.4e-4 at |shaping| = 3200 (a full-throttle lane at hundreds of m/s or rad/s;
# the yaw rate enters at 50 per rad/s and is not in the observation), which the fixed 5e-5 + 1e-5 |r| does not cover
# when the reward itself is small.  Added on top of it, per unit of the ORACLE's |shaping|: one ulp of the word (two
# where the state itself is float32-rounded).  Seeds 2102 and 2256 of the round-5 sweep: profiles/r05_fuzz_sweep.txt
REWARD_UNIT = {"float64": 0.0, "float32": 2.0 ** -23, "float32_rn": 2.0 ** -22}
HOVER = 0.016563        # per-motor value that just carries the DJI Phantom (tests/golden/meta.npz: hover_motor)


def draw_case(seed):
    rng = np.random.default_rng(1000 + seed)
    task = TASKS[seed % len(TASKS)]                     # every task appears
    cfg = dict(
        task=task,
        n=int(rng.choice([1, 63, 64, 65, 200, 517, 1024])),
        mode=str(rng.choice(["float32", "float32", "float64", "float32_rn"])),
        autoreset=str(rng.choice(["disabled", "next_step", "same_step"])),
        substeps=int(rng.choice([1, 1, 2, 3, 10])),
        seed=int(rng.integers(0, 2**31)),
        env_id_base=int(rng.choice([0, 64, 1000, 2**20 + 7])),
        time_limit_truncates=bool(rng.integers(0, 2)),
        episode_stats=bool(rng.integers(0, 2)),
    )
    kw = dict(
        max_steps=int(rng.choice([1000, 25, 40])),
        frames_per_second=int(rng.choice([100, 100, 50, 200])),
        initial_altitude=float(rng.choice([10.0, 3.0, 0.5])),
        bounds=float(rng.choice([10.0, 2.0])),
        initial_random_force=float(rng.choice([30.0, 0.0, 120.0])),
        max_angle=float(rng.choice([45.0, 10.0])),
    )
    if task.startswith("lander"):
        kw.update(target_radius=float(rng.choice([2.0, 0.5])), dz_max=float(rng.choice([10.0, 1.5])))
    law = str(rng.choice(["uniform", "near_hover", "descend", "mixed"]))
    return cfg, kw, law, rng


def draw_actions(rng, law, n, adim):
    if law == "uniform":
        a = rng.uniform(-1, 1, (n, adim))
    elif law == "near_hover":
        a = HOVER * (1 + 0.05 * rng.standard_normal((n, adim)))
    elif law == "descend":
        a = HOVER * (0.8 + 0.1 * rng.random((n, adim)))
    else:
        pick = rng.integers(0, 3, (n, 1))
        a = np.where(pick == 0, rng.uniform(-1, 1, (n, adim)),
                     np.where(pick == 1, HOVER * (1 + 0.05 * rng.standard_normal((n, adim))),
                              HOVER * 0.85 * np.ones((n, adim))))
    return a.astype(np.float32)


# the caller's OWN policy fused into the K-step kernel (gym_copter_amd.compile_policy): a replay policy -- the K x N
# action rows behind a one-float header holding N -- compiled once per (task, storage mode) of the cases that use it
_REPLAY_POLICY = """
struct Policy {
  const float* params;
  __device__ void load(uint32_t, bool) {}
  __device__ void store(uint32_t, bool) {}
  __device__ void operator()(const float (&)[OBS], uint32_t env, int k, bool, float (&a)[ACT]) const {
    const uint32_t n = (uint32_t)params[0];
    const float* row = params + 1 + ((size_t)k * n + (env < n ? env : 0u)) * ACT;
    for (int j = 0; j < ACT; ++j) a[j] = row[j];
  }
};
"""
_policies = {}
HAVE_HIPCC = bool(__import__("shutil").which("hipcc")) or __import__("os").path.exists("/opt/rocm/bin/hipcc")


def replay_policy(env, cache_dir):
    import gym_copter_amd
    key = (env.task, int(env.config.state_mode))
    if key not in _policies:
        _policies[key] = gym_copter_amd.compile_policy(env, _REPLAY_POLICY, cache_dir=cache_dir)
    return _policies[key]


@pytest.mark.parametrize("seed", range(64))
def test_random_configuration_and_stepping_forms_vs_oracle(seed, tmp_path_factory):
    run_case(seed, tmp_path_factory)


@pytest.mark.parametrize("seed", [2102, 2256])
def test_cases_whose_reward_shows_one_ulp_of_the_shaping_word(seed, tmp_path_factory):
    """Full-throttle lanes with |shaping| in the thousands and a small reward (REWARD_UNIT): the two seeds of the
    64..3199 sweep that the fixed reward limit refused although state, observation and flags agreed."""
    run_case(seed, tmp_path_factory)


# The two seeds of the 400-seed sweep that do NOT pass (see the header): chaos after a one-unit rounding difference, not a
# bug.  Kept as strict expected failures -- a change that makes one of them pass (or fail in another way: the second
# test pins HOW they fail) is a change in the device's rounding behaviour and must be noticed.
CHAOTIC = [(181, 2.7e-8), (360, 2.0e-7)]


@pytest.mark.xfail(strict=True, reason="known chaotic divergence after a one-unit difference of the stored format "
                                       "(auto-reset off, tumbling envs): tools/fuzz_sweep.py, rounds 3-5")
@pytest.mark.parametrize("seed,grown", CHAOTIC)
def test_known_chaotic_seeds_still_diverge(seed, grown, tmp_path_factory):
    run_case(seed, tmp_path_factory)


@pytest.mark.parametrize("seed,grown", CHAOTIC)
def test_known_chaotic_seeds_diverge_the_way_they_were_recorded(seed, grown, tmp_path_factory):
    """What the failure IS: a state (or float32 row) error just above the same-mode tolerance -- recorded 2.7e-8 /
    2.0e-7 -- and nothing discrete (flags, status, counters stay equal up to that point)."""
    import re
    with pytest.raises(AssertionError) as ei:
        run_case(seed, tmp_path_factory)
    msg = str(ei.value)
    m = re.search(r"(?:state|obs) err ([0-9.]+e[+-][0-9]+)", msg)
    assert m is not None, "seed %d fails in a NEW way (not a state / observation tolerance): %s" % (seed, msg[:300])
    err = float(m.group(1))
    assert grown / 5 <= err <= grown * 5, "seed %d: recorded growth %.1e, now %.3e" % (seed, grown, err)


def run_case(seed, tmp_path_factory):
    import torch
    cfg, kw, law, rng = draw_case(seed)
    track = bool(rng.integers(0, 2))
    env, orc = make_pair(cfg["task"], cfg["n"], cfg["mode"], autoreset=cfg["autoreset"], substeps=cfg["substeps"],
                         seed=cfg["seed"], env_id_base=cfg["env_id_base"],
                         time_limit_truncates=cfg["time_limit_truncates"], episode_stats=cfg["episode_stats"],
                         track_time=track, **kw)
    ctx = "seed %d %r %r law %s track %s" % (seed, cfg, kw, law, track)
    n, adim, tol = cfg["n"], env.action_dim, MODE_TOL[cfg["mode"]]
    obs0, _ = env.reset()
    want0 = orc.reset()
    assert np.allclose(to_np(obs0), want0, rtol=0, atol=1e-6), ctx
    served_ok = n <= env.serve_max_envs()
    t = 0
    while t < 90:
        forms = ["step", "many"] + (["served"] if served_ok else []) + (["policy"] if seed % 5 == 0 and HAVE_HIPCC else [])
        form = str(rng.choice(forms))
        k = int(rng.integers(1, 9))
        acts = np.stack([draw_actions(rng, law, n, adim) for _ in range(k)])
        if form == "step":
            outs = []
            for j in range(k):
                o, r, te, tr, _ = env.step(torch.from_numpy(acts[j]).to(env.device))
                outs.append(tuple(to_np(v).copy() for v in (o, r, te, tr)))
        elif form == "many":
            o, r, te, tr = env.step_many(torch.from_numpy(acts).to(env.device))
            o, r, te, tr = (to_np(v) for v in (o, r, te, tr))
            outs = [(o[j], r[j], te[j], tr[j]) for j in range(k)]
        elif form == "policy":
            pol = replay_policy(env, str(tmp_path_factory.getbasetemp() / "policies"))
            params = torch.from_numpy(np.concatenate([[np.float32(n)], acts.ravel()]).astype(np.float32)).to(env.device)
            o, r, te, tr, rec = (to_np(v) for v in env.rollout_policy(pol, k, params, return_actions=True))
            assert np.array_equal(rec, acts), ctx
            outs = [(o[j], r[j], te[j], tr[j]) for j in range(k)]
        else:
            env.serve_begin(k, ring=2, timeout=5.0)
            outs = []
            for j in range(k):
                env.serve_submit(j, torch.from_numpy(acts[j]).to(env.device))
                o, r, te, tr = env.serve_collect(j)
                outs.append(tuple(to_np(v).copy() for v in (o, r, te, tr)))
            assert env.serve_end() == k, ctx
        for j in range(k):
            sh0 = np.abs(np.nan_to_num(orc.prev_shaping.astype(np.float64)))
            want = orc.step(acts[j].astype(np.float64))
            sh = np.maximum(sh0, np.abs(np.nan_to_num(orc.prev_shaping.astype(np.float64))))
            # the float32 observation is a rounding of the stored word: where device and oracle differ by one unit of
            # the stored format (the state check below bounds that) the row may differ by one float32 ulp
            assert_step_close(outs[j], want, max(OBS_ULP, tol), ctx="%s form %s step %d" % (ctx, form, t + j),
                              r_unit=REWARD_UNIT[cfg["mode"]], shaping=sh)
        t += k
        assert_state_close(env, orc, tol, ctx="%s form %s after step %d" % (ctx, form, t))
        if cfg["autoreset"] == "disabled":
            # what a caller does without auto-reset: a masked reset of the envs that finished in this stretch (they
            # flew on for up to k - 1 steps past their end, as upstream lets them; left alone for the whole case
            # they tumble at hundreds of m/s, where one unit of the stored format grows past any fixed tolerance)
            m = np.zeros(n, bool)
            for o in outs:
                m |= o[2].astype(bool) | o[3].astype(bool)
            if m.any():
                obs_r, _ = env.reset(options={"mask": m})
                want_r = orc.reset(mask=m)
                assert np.allclose(to_np(obs_r)[m], want_r[m], rtol=0, atol=1e-6), ctx
                assert_state_close(env, orc, tol, ctx="%s after the masked reset at step %d" % (ctx, t))
        if track:
            assert np.array_equal(env.get_state(only=("ticks",))["ticks"], orc.ticks), "ticks %s step %d" % (ctx, t)
    if cfg["episode_stats"]:
        s = to_np(env.batch_stats())       # (envs, airborne, sum / max of the step counters, episodes, returns, nonfinite)
        assert s[0] == n and s[1] == np.sum(orc.status == 3) and s[2] == orc.steps.sum() and s[3] == orc.steps.max(), ctx
        assert s[6] == 0, ctx
    env.close()
