"""-m gpu: the HIP path against the CPU oracle run in the same storage mode (gpu_util.MODE_TOL: float64 rounding only) at
BASELINE's sizes and in randomised configurations -- storage modes and the stored-word codec, substeps, auto-reset modes,
ticks, counters, statistics, per-env vehicles, variants, perturbations, the NaN / inf guard, checkpoint round trips --
and size-independent properties (mirror symmetry, position invariance, neighbour independence)."""
import os

import numpy as np
import pytest

from conftest import load_cases
from gpu_util import (AUTORESET, MODE_TOL, VecOracle, assert_state_close, assert_step_close, have_gpu, make_pair,
                      scaled_err, step_both, to_np)

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not have_gpu(), reason="needs a HIP device")]

MODES = ["float32", "float32_rn", "float64"]
HOVER = float(np.load(os.path.join(os.path.dirname(__file__), "golden", "meta.npz"))["hover_motor"])
BAR = 1e-5   # BASELINE.json: <= 1e-5 relative fp32 per state component over 1000 steps
DYN = load_cases("dynamics_traces.npz")
# ---------------------------------------------------------------------------------------
# 1D / 2D task variants (motor fan-out + observation sub-selection)
# ---------------------------------------------------------------------------------------
VARIANTS = ["lander2d", "lander1d", "hover2d", "hover1d"]


# ---------------------------------------------------------------------------------------
# 1000 steps on identical motor inputs (the north-star parity statement)
# ---------------------------------------------------------------------------------------
def _rollout_vs_oracle(task, n, mode, T, law, substeps=1, check_every=100):
    rng = np.random.default_rng(2024)
    env, orc = make_pair(task, n, mode, substeps=substeps, seed=9)
    env.reset(seed=9)
    orc.reset(seed=9)
    assert_state_close(env, orc, 0.0, ctx="after reset")    # Philox forces identical
    st = env.get_state()
    assert np.array_equal(st["force"], orc.force.astype(np.float64))
    worst = 0.0
    for t in range(T):
        if law == "near_hover":
            a = (HOVER * (1 + 0.01 * rng.standard_normal((n, 4)))).astype(np.float32)
        elif law == "const":
            a = np.full((n, 4), 1.625e-2, dtype=np.float32)
        else:
            a = rng.uniform(-1, 1, (n, 4)).astype(np.float32)
        got, want, _ = step_both(env, orc, a)
        if t % check_every == 0 or t == T - 1:
            assert_step_close(got, want, MODE_TOL[mode], ctx="t=%d" % t)
            worst = max(worst, assert_state_close(env, orc, MODE_TOL[mode], ctx="t=%d" % t))
    env.close()
    return worst


# ---------------------------------------------------------------------------------------
# one step from random states: every output against the oracle in the same storage mode
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("task", ["lander3d", "hover3d"])
@pytest.mark.parametrize("mode", MODES)
def test_single_step_random_states(task, mode):
    rng = np.random.default_rng(11)
    n = 4096 + 37   # ragged last wavefront
    env, orc = make_pair(task, n, mode, seed=5)
    env.reset(options={"forces": np.zeros((3, n), np.float32)})
    orc.reset(forces=np.zeros((3, n)))
    x = rng.standard_normal((12, n)) * np.array([4, 2, 4, 2, 6, 2, .4, .5, .4, .5, 2, 1])[:, None]
    x[4] -= 6
    x[0, :64] = 9.99 + 0.02 * rng.random(64)          # bounds edge
    x[6, 64:128] = np.pi / 4 - 1e-3 + 2e-3 * rng.random(64)   # tilt edge
    x[4, 128:512] = np.abs(x[4, 128:512]) * 0.01       # below ground, some descending
    status = rng.integers(0, 4, n).astype(np.uint8)
    steps = rng.integers(1, 1002, n).astype(np.int32)
    steps[:16] = 1000
    prev = -rng.random(n) * 300
    prev[::97] = np.nan
    force = rng.uniform(-30, 30, (3, n))
    flags = (rng.random(n) < 0.3).astype(np.uint8)
    # make every input exactly representable in the storage mode under test
    orc.x[:] = orc._round(x)
    orc.status[:] = status
    orc.steps[:] = steps
    orc.prev_shaping[:] = prev.astype(orc.T)
    orc.force[:] = force.astype(orc.T)
    orc.pending[:] = flags.astype(bool)
    env.set_state(x=orc.x.astype(np.float64), status=status, steps=steps,
                  prev_shaping=orc.prev_shaping.astype(np.float64),
                  force=orc.force.astype(np.float64), flags=flags | 4)   # bit 2: install the given forces
    st = env.get_state()
    assert np.array_equal(st["x"], orc.x.astype(np.float64))      # set/get round trip is exact
    assert np.array_equal(np.isnan(st["prev_shaping"]), np.isnan(prev))
    actions = rng.uniform(-0.5, 1.5, (n, 4)).astype(np.float32)
    actions[::5] = (HOVER * (1 + 0.01 * rng.standard_normal((len(actions[::5]), 4)))).astype(np.float32)
    got, want, _ = step_both(env, orc, actions)
    # single step: states agree to float64 rounding (a handful of ulps of the stored word)
    tol = {"float64": 1e-13, "float32": 4e-9, "float32_rn": 2.5e-7}[mode]     # one unit of the stored last place
    assert_step_close(got, want, max(tol, 1.3e-7), ctx="%s %s" % (task, mode))
    assert_state_close(env, orc, tol, ctx="%s %s" % (task, mode))
    st = env.get_state()
    if task == "lander3d":
        ps, wps = st["prev_shaping"], orc.prev_shaping.astype(np.float64)
        assert np.all(np.abs(ps - wps) <= 3.1e-5 + 1e-12 * np.abs(wps))
    env.close()


@pytest.mark.parametrize("mode", MODES)
def test_1000_steps_near_hover_vs_oracle(mode):
    worst = _rollout_vs_oracle("lander3d", 2048, mode, 1000, "near_hover")
    print("1000-step near-hover, %s: worst scaled state error vs same-mode oracle %.3e" % (mode, worst))


def test_1000_steps_vs_float64_reference_semantics():
    """Default device format (float32 words + guard bits) against the oracle run in pure
    float64 (= the reference's arithmetic, pinned to its golden traces): <= 1e-5 * max(|ref|,1)
    on every state component at every 50th step of 1000, constant and near-hover thrust."""
    rng = np.random.default_rng(7)
    n = 1024
    for law in ("const", "near_hover"):
        env, _ = make_pair("lander3d", n, "float32", seed=3)
        _, ref = make_pair("lander3d", n, "float64", seed=3)
        _.close()
        env.reset(seed=3)
        ref.reset(seed=3)
        # the float64 oracle must start from the float32-rounded forces the device holds
        ref.force[:] = ref.force.astype(np.float32).astype(np.float64)
        worst = 0.0
        for t in range(1000):
            a = (np.full((n, 4), 1.625e-2) if law == "const"
                 else HOVER * (1 + 0.01 * rng.standard_normal((n, 4)))).astype(np.float32)
            got, want, _i = step_both(env, ref, a)
            if t % 50 == 0 or t == 999:
                st = env.get_state()
                airborne = ref.status == 3    # grounded copters are frozen; compare the flying ones
                e = scaled_err(st["x"][:, airborne], ref.x[:, airborne])
                worst = max(worst, e)
                assert e <= BAR, (law, t, e)
        print("1000 steps vs float64 semantics [%s]: worst %.3e (bar %.0e)" % (law, worst, BAR))
        env.close()


@pytest.mark.parametrize("task,n", [("lander3d", 65536), ("hover3d", 262144)])
def test_full_size_short_rollout_vs_oracle(task, n):
    """BASELINE configs 2 and 3 at full batch size, random U[-1,1) actions with NEXT_STEP
    auto-reset (reset churn ~15 % of lanes per step): 40 steps against the oracle, every
    output of every step."""
    rng = np.random.default_rng(99)
    env, orc = make_pair(task, n, "float32", autoreset="next_step", seed=1234)
    env.reset()
    orc.reset()
    resets = 0
    for t in range(40):
        a = rng.uniform(-1, 1, (n, 4)).astype(np.float32)
        got, want, _ = step_both(env, orc, a)
        # diverging (full-throttle) trajectories: compare at the float32-observation level
        assert_step_close(got, want, 2e-6, r_abs="auto", ctx="%s t=%d" % (task, t))
        resets += int(want[2].sum())
    assert_state_close(env, orc, 2e-6, ctx=task)
    assert resets > n        # every env finished more than once on average
    env.close()


@pytest.mark.parametrize("case", range(16))
def test_randomised_task_parameters_vs_oracle(case):
    """Every constructor keyword of the task at once, drawn at random per case -- _Task's keywords
    (task.py:32-38), Lander's class constants (lander.py:17-23), the frame rate, the step limit, substeps --
    with a mixed action law and auto-reset churn: 300 steps of 256 envs against the oracle, both tasks, all
    three auto-reset modes, all three state-word modes."""
    rng = np.random.default_rng(5000 + case)
    task = ("lander3d", "hover3d")[case % 2]
    autoreset = ("next_step", "same_step", "disabled")[case % 3]
    kw = dict(initial_random_force=float(rng.uniform(0, 60)), out_of_bounds_penalty=float(rng.uniform(10, 300)),
              max_angle=float(rng.uniform(20, 70)), bounds=float(rng.uniform(3, 20)),
              initial_altitude=float(rng.uniform(0.5, 12)), max_steps=int(rng.integers(20, 400)),
              frames_per_second=int(rng.choice([50, 100, 200])),
              target_radius=float(rng.uniform(0.5, 5)), yaw_penalty_factor=float(rng.uniform(0, 100)),
              xyz_penalty_factor=float(rng.uniform(1, 60)), dz_max=float(rng.uniform(1, 15)),
              dz_penalty=float(rng.uniform(0, 200)), inside_radius_bonus=float(rng.uniform(0, 300)))
    mode = ("float32", "float64", "float32_rn")[(case // 2) % 3]
    n, T = 256, 300
    seed = (case, 2 ** 63 + case, 2 ** 64 - 1 - case)[int(rng.integers(3))]     # every bit of the seed matters
    base = (0, 123456789, 2 ** 32 - n)[int(rng.integers(3))]                     # ... and of the global env id
    env, orc = make_pair(task, n, mode, autoreset=autoreset, seed=seed, substeps=int(rng.choice([1, 1, 3])),
                         time_limit_truncates=bool(case & 4), env_id_base=base, **kw)
    assert float(env.config.target_radius) == kw["target_radius"] and env.config.max_steps == kw["max_steps"]
    env.reset()
    orc.reset()
    ends = 0
    for t in range(T):
        a = np.empty((n, 4), dtype=np.float32)
        a[0::2] = rng.uniform(-1, 1, (n // 2, 4))
        a[1::2] = HOVER * rng.uniform(0.97, 1.01) * (1 + 0.01 * rng.standard_normal((n // 2, 4)))
        got, want, _ = step_both(env, orc, a)
        assert_step_close(got, want, 2e-6, r_abs="auto", ctx="case %d %s %s t=%d" % (case, task, autoreset, t))
        done = want[2] | want[3]
        ends += int(done.sum())
        if autoreset == "disabled" and done.any():      # the caller resets what finished (masked reset)
            env.reset(options={"mask": done})
            orc.reset(mask=done)
    assert_state_close(env, orc, 2e-6, ctx="case %d" % case)
    assert ends > 0
    env.close()


def test_long_soak_mixed_actions_vs_oracle():
    """20 000 steps of 384 envs (7.7 M env-steps, thousands of episodes per env: the episode counters, the
    Philox draws keyed by them and the step-limit path all run far past anything a short test reaches) with a
    mixed action law -- a third of the envs random (crash / tilt / out-of-bounds churn), a third near hover
    (episodes end at the 1000-step limit), a third descending gently from 2 m (soft landings with the bonus, and
    crashes when the reset perturbation pushes the sink rate past the limit); every output of every step against
    the oracle, and the full state at the end."""
    import torch
    n, T = 384, 20000
    rng = np.random.default_rng(2024)
    env, orc = make_pair("lander3d", n, "float32", autoreset="next_step", seed=99, initial_altitude=2.0)
    env.reset()
    orc.reset()
    hover = HOVER
    ends = np.zeros(n, dtype=np.int64)
    bonus = 0
    seen = set()
    chunk = 500
    for t0 in range(0, T, chunk):
        a = np.empty((chunk, n, 4), dtype=np.float32)
        a[:, 0::3] = rng.uniform(-1, 1, (chunk, n // 3, 4))
        a[:, 1::3] = hover * (1 + 0.01 * rng.standard_normal((chunk, n // 3, 4)))
        a[:, 2::3] = 0.99 * hover * (1 + 0.002 * rng.standard_normal((chunk, n // 3, 4)))   # -0.2 m/s^2
        for k in range(chunk):
            got, want, _ = step_both(env, orc, a[k])
            assert_step_close(got, want, 2e-6, r_abs="auto", ctx="t=%d" % (t0 + k))
            # the near-hover third does not diverge: its rewards get the tight bound of the golden-trace tests
            # (the tolerance above widens with an env's magnitude, for the full-throttle third), leaving out the steps
            # around an episode's end
            calm = ~(want[2][1::3] | got[2][1::3].astype(bool))
            dr = np.abs(got[1][1::3].astype(np.float64) - want[1][1::3])[calm]
            assert dr.size == 0 or dr.max() <= 5e-5, ("near-hover reward", t0 + k, float(dr.max()))
            ends += want[2]
            bonus += int((want[1][2::3] > 50).sum())
            seen |= set(np.unique(orc.status).tolist())
    assert_state_close(env, orc, 2e-6, ctx="soak")
    assert ends[0::3].min() > 500 and ends[1::3].min() >= 15 and ends[2::3].min() >= 15, (
        ends[0::3].min(), ends[1::3].min(), ends[2::3].min())
    assert bonus > 1000 and seen == {0, 1, 2, 3}       # soft landings with the bonus; every flight status met
    st = env.get_state()
    assert st["episode"].max() > 1000
    env.close()


def test_substeps_config5():
    """BASELINE config 5: dt = 1e-3 with 10 inner substeps per step()."""
    g = DYN["D10_fps1000"]
    env, orc = make_pair("lander3d", 256, "float64", substeps=10)
    f = np.tile(g["force"][:3, None], (1, 256)).astype(np.float32)
    env.reset(options={"forces": f})
    orc.reset(forces=f.astype(np.float64))
    import torch
    a = torch.from_numpy(np.tile(g["motors"][0].astype(np.float32), (256, 1))).to(env.device)
    for s in range(1000):
        env.step(a)
        if s % 100 == 99:
            st = env.get_state()
            assert scaled_err(st["x"][:, 7], g["x"][10 * s + 9]) <= 1e-10, s
    env.close()
    worst = _rollout_vs_oracle("lander3d", 1024, "float32", 300, "near_hover", substeps=10)
    assert worst <= MODE_TOL["float32"]


# ---------------------------------------------------------------------------------------
# auto-reset, done-list compaction, episode statistics
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("task", ["lander3d", "hover3d"])
@pytest.mark.parametrize("autoreset", ["next_step", "same_step"])
def test_autoreset_and_done_list(task, autoreset):
    rng = np.random.default_rng(4)
    n = 3000
    env, orc = make_pair(task, n, "float32", autoreset=autoreset, seed=77, env_id_base=10 ** 6,
                         episode_stats=True)
    env.enable_done_list()
    if autoreset == "same_step":
        env.enable_final_obs()
    env.reset()
    orc.reset()
    total = 0
    for t in range(60):
        a = rng.uniform(-1, 1, (n, 4)).astype(np.float32)
        got, want, infos = step_both(env, orc, a)
        assert_step_close(got, want, 2e-6, r_abs="auto", ctx="t=%d" % t)
        ep = infos["episode"]
        cnt = int(to_np(ep["count"])[0])
        ids = to_np(ep["ids"])[:cnt]
        order = np.argsort(ids)
        assert np.array_equal(ids[order], orc.last_done), t
        assert np.array_equal(to_np(ep["length"])[:cnt][order], orc.last_length), t
        assert np.allclose(to_np(ep["return"])[:cnt][order], orc.last_return, rtol=1e-5, atol=1e-2), t
        if autoreset == "same_step" and cnt:
            fo = to_np(infos["final_obs"])[orc.last_done]
            assert scaled_err(fo, orc.final_obs[orc.last_done]) <= 2e-6
        total += cnt
    assert total > n
    assert_state_close(env, orc, 2e-6)
    st = env.get_state()
    assert np.array_equal(st["force"], orc.force.astype(np.float64))   # Philox draws identical
    assert np.array_equal(st["episode"], orc.episode)
    env.close()


def test_time_limit_as_truncation():
    env, orc = make_pair("hover3d", 130, "float32", time_limit_truncates=True, max_steps=25)
    z = np.zeros((3, 130), np.float32)
    env.reset(options={"forces": z})
    orc.reset(forces=z)
    a = np.full((130, 4), HOVER, dtype=np.float32)
    import torch
    many, _ = make_pair("hover3d", 130, "float32", time_limit_truncates=True, max_steps=25)
    many.reset(options={"forces": z})
    obs_m, rew_m, term_m, trunc_m = many.step_many(torch.from_numpy(np.broadcast_to(a, (25, 130, 4)).copy()).to(many.device))
    for t in range(25):
        got, want, _ = step_both(env, orc, a)
        assert_step_close(got, want, 1e-7)
        assert got[3].all() == (t == 24) and not got[2].any()
        assert np.array_equal(to_np(trunc_m[t]), got[3]) and np.array_equal(to_np(obs_m[t]), got[0])   # K-step kernel too
    env.close()
    many.close()


# ---------------------------------------------------------------------------------------
# size-independent properties at full size; edge cases
# ---------------------------------------------------------------------------------------
def test_batch_position_invariance_full_size():
    """Env i's trajectory does not depend on batch size, position in the batch or shard:
    the same global env ids stepped inside a 65 536 batch and as a 1000-env shard."""
    import gym_copter_amd
    import torch
    rng = np.random.default_rng(1)
    N, lo, m = 65536, 31337, 1000
    big = gym_copter_amd.CopterVecEnv("lander3d", N, seed=42, autoreset_mode="next_step")
    small = gym_copter_amd.CopterVecEnv("lander3d", m, seed=42, autoreset_mode="next_step",
                                        env_id_base=lo)
    big.reset()
    small.reset()
    for t in range(100):
        a = (HOVER * (1 + 0.3 * rng.standard_normal((N, 4)))).astype(np.float32)
        ob, rb, tb, _, _ = big.step(torch.from_numpy(a).to(big.device))
        os_, rs, ts, _, _ = small.step(torch.from_numpy(a[lo:lo + m]).to(small.device))
        assert torch.equal(ob[lo:lo + m], os_) and torch.equal(rb[lo:lo + m], rs)
        assert torch.equal(tb[lo:lo + m], ts)
    sb, ss = big.get_state(), small.get_state()
    assert np.array_equal(sb["x"][:, lo:lo + m], ss["x"]) and np.array_equal(sb["steps"][lo:lo + m], ss["steps"])
    big.close()
    small.close()


@pytest.mark.parametrize("n", [1, 63, 64, 65, 255, 257])
def test_ragged_batch_sizes(n):
    rng = np.random.default_rng(n)
    for task in ("lander3d", "hover3d"):
        env, orc = make_pair(task, n, "float32", autoreset="next_step", seed=n)
        env.reset()
        orc.reset()
        for t in range(30):
            got, want, _ = step_both(env, orc, rng.uniform(-1, 1, (n, 4)).astype(np.float32))
            assert_step_close(got, want, 2e-6, r_abs="auto")
        env.close()


def test_nonfinite_and_out_of_range_inputs_propagate_like_the_reference():
    """The reference raises nothing on the path: NaN/inf propagate silently, actions are
    clipped to [0,1] (task.py:91)."""
    env, orc = make_pair("lander3d", 64, "float64")
    z = np.zeros((3, 64), np.float32)
    env.reset(options={"forces": z})
    orc.reset(forces=z)
    a = np.full((64, 4), HOVER, dtype=np.float32)
    a[0] = [5.0, -3.0, 1e30, -1e30]
    a[1, 2] = np.nan
    a[2] = np.inf
    for t in range(3):
        got, want, _ = step_both(env, orc, a)
    st = env.get_state()
    assert scaled_err(st["x"][:, 0], orc.x[:, 0]) < 1e-12
    assert np.isnan(st["x"][:, 1]).any() == np.isnan(orc.x[:, 1]).any()
    assert scaled_err(st["x"][:, 3:], orc.x[:, 3:]) < 1e-12
    env.close()


@pytest.mark.parametrize("autoreset", ["next_step", "same_step", "disabled"])
@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("task", VARIANTS)
def test_variants_match_oracle_and_step_many(task, mode, autoreset):
    """Random and near-hover actions on a ragged batch of a 1D / 2D variant: every output of every
    step against the oracle in the same storage mode, and cs_step_many bit-identical to the single
    steps."""
    import torch
    rng = np.random.default_rng(23)
    n, K = 3001, 30
    env, orc = make_pair(task, n, mode, autoreset=autoreset, seed=3, episode_stats=True)
    many, _ = make_pair(task, n, mode, autoreset=autoreset, seed=3, episode_stats=True)
    A = env.action_dim
    assert env.single_action_space.shape == (A,) and env.single_observation_space.shape == (env.obs_dim,)
    assert np.array_equal(to_np(env.reset()[0]), orc.reset())
    many.reset()
    tol = MODE_TOL[mode]
    for chunk in range(3):
        law = rng.uniform(-1, 1, (K, n, A)) if chunk != 1 else HOVER * (1 + 0.05 * rng.standard_normal((K, n, A)))
        acts = law.astype(np.float32)
        obs_m, rew_m, term_m, trunc_m = many.step_many(torch.from_numpy(acts).to(many.device))
        for k in range(K):
            got, want, _ = step_both(env, orc, acts[k])
            assert_step_close(got, want, max(tol, 2e-6), r_abs="auto", ctx=(task, mode, chunk, k))
            assert np.array_equal(to_np(obs_m[k]), got[0]) and np.array_equal(to_np(rew_m[k]), got[1])
            assert np.array_equal(to_np(term_m[k]), got[2]) and np.array_equal(to_np(trunc_m[k]), got[3])
        assert_state_close(env, orc, max(tol, 2e-6))
    env.close()
    many.close()


@pytest.mark.parametrize("task,mode,autoreset", [("lander3d", "float32", "next_step"),
                                                 ("hover3d", "float64", "same_step"),
                                                 ("lander2d", "float32_rn", "disabled")])
def test_randomised_vehicles_match_oracle(task, mode, autoreset):
    """Domain randomisation: every env of a ragged batch gets its own vehicle (+-30 % around the DJI
    Phantom) and gravity (Mars .. 1.2 g); single steps, cs_step_many and cs_set_motors against the
    oracle run with the same per-env parameter arrays."""
    import torch
    from oracle.refcpu import DJI_PHANTOM, VehicleParams
    rng = np.random.default_rng(31)
    n, K = 2777, 25
    base = np.array([getattr(DJI_PHANTOM, k) for k in ("B", "D", "M", "L", "Ix", "Iy", "Iz", "Jr", "maxrpm")] + [9.80665])
    table = base[:, None] * rng.uniform(0.7, 1.3, (10, n))
    table[9] = rng.uniform(3.7, 11.8, n)
    env, _ = make_pair(task, n, mode, autoreset=autoreset, seed=13, episode_stats=True)
    many, _ = make_pair(task, n, mode, autoreset=autoreset, seed=13, episode_stats=True)
    orc = VecOracle(task, n, substeps=1, store_mode=mode, autoreset=AUTORESET[autoreset], seed=13,
                    vp=VehicleParams(*[table[j].copy() for j in range(9)]), g=table[9].copy())
    for e in (env, many):
        e.set_vehicle_params(table)
    assert np.array_equal(to_np(env.reset()[0]), orc.reset())
    many.reset()
    A = env.action_dim
    hover = np.sqrt(table[2] * table[9] / (4 * table[0])) / (table[8] * np.pi / 30)      # per-env hover motor value
    tol = max(MODE_TOL[mode], 2e-6)
    for chunk in range(3):
        law = rng.uniform(-1, 1, (K, n, A)) if chunk == 0 else hover[None, :, None] * (1 + 0.05 * rng.standard_normal((K, n, A)))
        acts = law.astype(np.float32)
        obs_m, rew_m, term_m, trunc_m = many.step_many(torch.from_numpy(acts).to(many.device))
        for k in range(K):
            got, want, _ = step_both(env, orc, acts[k])
            assert_step_close(got, want, tol, r_abs="auto", ctx=(task, mode, chunk, k))
            assert np.array_equal(to_np(obs_m[k]), got[0]) and np.array_equal(to_np(rew_m[k]), got[1])
            assert np.array_equal(to_np(term_m[k]), got[2])
        assert_state_close(env, orc, tol)
    # on-device random policy with per-env parameters: the rollout's actions replayed as single steps
    obs_r, rew_r, term_r, trunc_r, act_r = many.rollout_random(6, return_actions=True)
    for k in range(6):
        got, want, _ = step_both(env, orc, to_np(act_r[k]))
        assert np.array_equal(to_np(obs_r[k]), got[0]) and np.array_equal(to_np(rew_r[k]), got[1]), k
        assert_step_close(got, want, tol, r_abs="auto", ctx=("rollout", k))
    # dynamics-only entry point with per-env parameters
    m = rng.uniform(0, 0.05, (n, 4)).astype(np.float32)
    env.set_motors(torch.from_numpy(m).to(env.device))
    orc.set_motors(m.astype(np.float64))
    assert_state_close(env, orc, tol)
    # back to the uniform vehicle: same as a fresh env
    env.set_vehicle_params(None)
    fresh, _ = make_pair(task, n, mode, autoreset=autoreset, seed=99)
    f = rng.uniform(-30, 30, (3, n)).astype(np.float32)     # (the Philox draw depends on the episode count)
    env.reset(options={"forces": f})
    fresh.reset(options={"forces": f})
    a = rng.uniform(-1, 1, (n, A)).astype(np.float32)
    r1 = env.step(torch.from_numpy(a).to(env.device))
    r2 = fresh.step(torch.from_numpy(a).to(env.device))
    assert torch.equal(r1[0], r2[0]) and torch.equal(r1[1], r2[1])
    for e in (env, many, fresh):
        e.close()


def test_set_perturbation_mid_flight():
    """Dynamics.perturb() for the batch: a force installed between steps enters the next
    integrating call (twice, as upstream applies it) and is then gone."""
    rng = np.random.default_rng(5)
    n = 777
    env, orc = make_pair("lander3d", n, "float32", seed=2)
    env.reset()
    orc.reset()
    a = (HOVER * (1 + 0.01 * rng.standard_normal((n, 4)))).astype(np.float32)
    for t in range(3):
        got, want, _ = step_both(env, orc, a)
    f = rng.uniform(-50, 50, (3, n)).astype(np.float32).astype(np.float64)
    env.set_perturbation(f)
    orc.force[:] = f.astype(orc.T)
    orc.pending[:] = True
    dx_before = env.get_state()["x"][1].copy()
    got, want, _ = step_both(env, orc, a)
    assert_step_close(got, want, 2e-6, r_abs="auto")
    kick = env.get_state()["x"][1] - dx_before
    assert np.allclose(kick, 2 * f[0] / 1.380 * 0.01, rtol=0, atol=2e-3)     # 2 F/M dt on top of the thrust term
    got, want, _ = step_both(env, orc, a)
    assert_step_close(got, want, 2e-6, r_abs="auto")
    assert_state_close(env, orc, 2e-6)
    env.close()


@pytest.mark.parametrize("mode", MODES)
def test_export_state_on_device(mode):
    """cs_export_state (Dynamics.getState / getStatus as device tensors) agrees with the host-side
    cs_get_state and with the observation the step returned."""
    rng = np.random.default_rng(2)
    n = 3001
    env, _ = make_pair("lander3d", n, mode, autoreset="next_step", seed=4)
    env.reset()
    import torch
    for t in range(12):
        obs, *_ = env.step(torch.from_numpy(rng.uniform(-1, 1, (n, 4)).astype(np.float32)).to(env.device))
    st, dev = env.get_state(), env.state_tensors()
    with np.errstate(over="ignore"):
        assert np.array_equal(to_np(dev["x"]), st["x"].astype(np.float32))
    assert np.array_equal(to_np(dev["status"]), st["status"]) and np.array_equal(to_np(dev["steps"]), st["steps"])
    live = (st["flags"] & 2) == 0             # envs waiting for their reset return the finished state's observation too
    assert np.array_equal(to_np(dev["x"])[:10].T[live], to_np(obs)[live])
    env.close()


@pytest.mark.parametrize("task,mode", [("lander3d", "float32"), ("hover3d", "float32_rn"), ("lander2d", "float64")])
def test_random_pose_resets_match_oracle(task, mode):
    """Masked resets to random poses (some on the ground, some past the tilt limit), with and without
    the perturbation, interleaved with steps: device vs oracle, state for state."""
    rng = np.random.default_rng(77)
    n = 1500
    env, orc = make_pair(task, n, mode, autoreset="disabled", seed=12)
    env.reset()
    orc.reset()
    A = env.action_dim
    for rnd in range(4):
        poses = np.stack([rng.uniform(-9, 9, n), rng.uniform(-9, 9, n), rng.uniform(0, 12, n),
                          rng.uniform(-50, 50, n), rng.uniform(-50, 50, n)]).astype(np.float32)
        poses[2, ::7] = 0.0                         # on the ground
        mask = rng.random(n) < 0.6
        perturb = rnd % 2 == 0
        obs, _ = env.reset(options={"pose": poses, "mask": mask, "perturb": perturb})
        want = orc.reset(mask=mask, poses=poses.astype(np.float64), perturb=perturb)
        assert scaled_err(to_np(obs), want) <= (1e-7 if mode != "float64" else 0)
        assert_state_close(env, orc, max(MODE_TOL[mode], 1e-7))
        for t in range(6):
            a = (HOVER * (1 + 0.2 * rng.standard_normal((n, A)))).astype(np.float32)
            got, want, _ = step_both(env, orc, a)
            assert_step_close(got, want, max(MODE_TOL[mode], 2e-6), r_abs="auto", ctx=(rnd, t))
    env.close()


def test_float32_motor_model_vs_scalar_oracle_batch():
    """The float32 motor model over a batch with reset churn against the scalar oracle's passthrough
    mode (real NumPy float32 arithmetic), env by env."""
    import torch
    from oracle.refcpu import TaskOracle
    n, T = 24, 120
    rng = np.random.default_rng(3)
    env, _ = make_pair("lander3d", n, "float64", action_arith="float32")
    forces = rng.uniform(-30, 30, (3, n)).astype(np.float32)
    env.reset(options={"forces": forces})
    orcs = [TaskOracle("lander3d", action_dtype_passthrough=True) for _ in range(n)]
    for i, o in enumerate(orcs):
        o.reset(force_xyz=forces[:, i].astype(np.float64))
    acts = (HOVER * (1 + 0.2 * rng.standard_normal((T, n, 4)))).astype(np.float32)
    for t in range(T):
        obs, r, term, _, _ = env.step(torch.from_numpy(acts[t]).to(env.device))
        obs, r, term = to_np(obs), to_np(r), to_np(term)
        for i, o in enumerate(orcs):
            wobs, wr, wdone, _, _ = o.step(acts[t, i])
            assert scaled_err(obs[i], wobs) <= 1e-9, (t, i)
            assert bool(term[i]) == bool(wdone) and abs(float(r[i]) - wr) <= 5e-5 + 1e-5 * abs(wr), (t, i)
    env.close()


def test_config5_ten_substeps_at_65536_envs_vs_oracle():
    """BASELINE config 5 at its real size: Lander3D, 65 536 envs, dt = 1e-3 x 10 Dynamics.setMotors calls
    per step, near-hover actions, 100 steps, every 20th step and the final state against the oracle."""
    n, T = 65536, 100
    rng = np.random.default_rng(11)
    env, orc = make_pair("lander3d", n, "float32", substeps=10, seed=4)
    env.reset()
    orc.reset()
    for t in range(T):
        a = (HOVER * (1 + 0.01 * rng.standard_normal((n, 4)))).astype(np.float32)
        got, want, _ = step_both(env, orc, a)
        if t % 20 == 0 or t == T - 1:
            assert_step_close(got, want, MODE_TOL["float32"], ctx="t=%d" % t)
    worst = assert_state_close(env, orc, MODE_TOL["float32"])
    print("config 5 at 65 536 envs: worst scaled state error vs the oracle after %d steps %.3e" % (T, worst))
    env.close()


@pytest.mark.parametrize("mode,autoreset", [("float32", "next_step"), ("float64", "same_step")])
def test_mars_model_steps_match_oracle(mode, autoreset):
    """The full step (task logic, auto-reset, cs_step_many) on the Mars model, uniform Ingenuity
    parameters from cs_config, against the oracle with the same model."""
    import torch
    import gym_copter_amd
    from gpu_util import AUTORESET
    from oracle.refcpu import TaskParams, VehicleParams
    from oracle.refvec import VecOracle
    n, K = 1500, 20
    rng = np.random.default_rng(8)
    vp = dict(B=5.e-6, D=2.e-6, M=1.380, L=0.350, Ix=2, Iy=2, Iz=3, Jr=38e-4, maxrpm=15000)
    kw = dict(task="lander3d", num_envs=n, state_dtype=mode, autoreset_mode=autoreset, seed=6,
              vehicle_params=dict(vp, C_L=0.4), world_params=dict(G=3.721, rho=0.017), thrust_model="lift",
              rotor_gyro=True)
    env, many = gym_copter_amd.CopterVecEnv(**kw), gym_copter_amd.CopterVecEnv(**kw)
    orc = VecOracle("lander3d", n, TaskParams(), vp=VehicleParams(**vp), store_mode=mode,
                    autoreset=AUTORESET[autoreset], seed=6, g=3.721, mars=(0.017, 0.4))
    assert np.array_equal(to_np(env.reset()[0]), orc.reset())
    many.reset()
    hover = 0.26717326
    tol = max(MODE_TOL[mode], 2e-6)
    for chunk in range(3):
        law = rng.uniform(-1, 1, (K, n, 4)) if chunk == 0 else hover * (1 + 0.05 * rng.standard_normal((K, n, 4)))
        acts = law.astype(np.float32)
        obs_m, rew_m, term_m, _ = many.step_many(torch.from_numpy(acts).to(many.device))
        for k in range(K):
            got, want, _ = step_both(env, orc, acts[k])
            assert_step_close(got, want, tol, r_abs="auto", ctx=(mode, chunk, k))
            assert np.array_equal(to_np(obs_m[k]), got[0]) and np.array_equal(to_np(rew_m[k]), got[1])
            assert np.array_equal(to_np(term_m[k]), got[2])
        assert_state_close(env, orc, tol)
    env.close()
    many.close()


# ---------------------------------------------------------------------------------------
# device-side Dynamics.perturb (masked) and batch statistics
# ---------------------------------------------------------------------------------------
def test_set_perturbation_is_masked_and_graph_capturable():
    import torch
    n = 777
    rng = np.random.default_rng(1)
    env, orc = make_pair("lander3d", n, "float32", seed=3)
    env.reset()
    orc.reset()
    a = (HOVER * np.ones((n, 4))).astype(np.float32)
    for _ in range(3):
        step_both(env, orc, a)                       # the reset perturbation is consumed
    f = rng.uniform(-20, 20, (3, n)).astype(np.float32)
    mask = rng.random(n) < 0.4
    ft, mt = torch.from_numpy(f).to(env.device), torch.from_numpy(mask).to(env.device)
    g = torch.cuda.CUDAGraph()                       # cs_set_perturbation only enqueues: capturable
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        env.set_perturbation(ft, mask=mt)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        env.set_perturbation(ft, mask=mt)
    g.replay()
    orc.force[:, mask] = f[:, mask].astype(orc.T)
    orc.pending[mask] = True
    st = env.get_state()
    assert np.array_equal((st["flags"] & 1).astype(bool), mask)
    assert np.array_equal((st["flags"] & 4) != 0, mask)            # explicitly installed forces
    assert np.array_equal(st["force"][:, mask], f[:, mask].astype(np.float64))
    for _ in range(3):
        got, want, _ = step_both(env, orc, a)
        assert_step_close(got, want, MODE_TOL["float32"])
    assert_state_close(env, orc, MODE_TOL["float32"])
    env.close()


def test_episode_stats_match_the_state():
    n = 5000
    rng = np.random.default_rng(2)
    env, _ = make_pair("hover3d", n, "float32", autoreset="next_step", seed=1, episode_stats=True)
    env.reset()
    import torch
    for _ in range(25):
        env.step(torch.from_numpy(rng.uniform(-1, 1, (n, 4)).astype(np.float32)).to(env.device))
    s = to_np(env.batch_stats())
    st = env.get_state()
    assert s[0] == n and s[1] == np.sum(st["status"] == 3)
    assert s[2] == st["steps"].sum() and s[3] == st["steps"].max() and s[4] == st["episode"].sum()
    assert abs(s[5] - st["episode_return"].sum()) <= 1e-6 * max(1.0, abs(s[5]))
    env.close()


def test_float64_mode_at_the_context_size_limit():
    """2^25 envs in float64 words: 5.8 GB of tiles, byte offsets past 2^32 (the tile base is 64-bit).
    Every env gets the same inputs, so the last tile must equal the first after reset + steps."""
    import torch
    import gym_copter_amd
    n = 1 << 25
    env = gym_copter_amd.CopterVecEnv("lander3d", n, state_dtype="float64", seed=0)
    z = torch.zeros((3, n), dtype=torch.float32, device=env.device)
    z[0] = 7.5
    env.reset(options={"forces": z})
    a = torch.full((n, 4), 1.7e-2, dtype=torch.float32, device=env.device)
    for _ in range(3):
        obs, r, term, _, _ = env.step(a)
    assert torch.equal(obs[:64], obs[-64:]) and torch.equal(obs[0], obs[n // 2 + 12345])
    assert float(obs[0, 1]) != 0.0 and float(obs[-1, 4]) < -9.9
    t = env.state_tensors()
    assert torch.equal(t["x"][:, :64], t["x"][:, -64:]) and int(t["steps"][-1]) == 4
    env.close()


@pytest.mark.parametrize("task,n", [("lander3d", 65536), ("hover3d", 262144)])
def test_mirror_symmetry_at_full_size(task, n):
    """A property of the rigid body that needs no oracle, at BASELINE's full batch sizes: reflect the world in
    the x-z plane (y, dy, roll, roll rate, yaw, yaw rate and the lateral perturbation change sign; the motors
    swap 0<->2 and 1<->3, which negates the roll and yaw torques and keeps thrust and pitch torque,
    dynamics/__init__.py:127-132, :231-247) and the trajectory is the reflected trajectory: same rewards, same
    terminations, mirrored observations -- to float64 rounding (the sums of the motor model are re-associated
    by the swap)."""
    import torch
    import gym_copter_amd
    g = torch.Generator(device="cuda")
    g.manual_seed(17)
    a_env = gym_copter_amd.CopterVecEnv(task, n, state_dtype="float64", autoreset_mode="disabled")
    b_env = gym_copter_amd.CopterVecEnv(task, n, state_dtype="float64", autoreset_mode="disabled")
    forces = (torch.rand((3, n), generator=g, device="cuda") * 2 - 1) * 30
    mirrored = forces.clone()
    mirrored[1] = -mirrored[1]
    oa, _ = a_env.reset(options={"forces": forces})
    ob, _ = b_env.reset(options={"forces": mirrored})
    od = a_env.obs_dim
    sign = torch.ones(od, device="cuda")
    for slot in (2, 3, 6, 7, 10, 11):          # y, dy, phi, dphi, psi, dpsi
        if slot < od:
            sign[slot] = -1
    assert torch.equal(oa * sign, ob)
    swap = torch.tensor([2, 3, 0, 1], device="cuda")
    ends = 0
    for t in range(60):
        a = torch.rand((n, 4), generator=g, device="cuda") * 0.05       # around hover thrust: long flights
        if t % 3 == 0:
            a = torch.rand((n, 4), generator=g, device="cuda") * 2 - 1     # and violent ones
        ra = [x.clone() for x in a_env.step(a)[:4]]
        rb = b_env.step(a[:, swap].contiguous())[:4]
        err = ((ra[0] * sign - rb[0]).abs() / rb[0].abs().clamp(min=1.0)).max().item()
        assert err <= 1e-6, (t, err)       # float32 observations of float64 states
        assert torch.equal(ra[2], rb[2]) and torch.equal(ra[3], rb[3]), t
        assert ((ra[1] - rb[1]).abs() <= 1e-3 + 1e-6 * rb[1].abs()).all(), t
        ends += int(ra[2].sum())
    sa, sb = a_env.get_state(), b_env.get_state()
    xs = np.ones(12)
    xs[[2, 3, 6, 7, 10, 11]] = -1
    assert scaled_err(sa["x"] * xs[:, None], sb["x"]) <= 1e-9
    assert np.array_equal(sa["status"], sb["status"]) and ends > 0
    a_env.close()
    b_env.close()


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


# ---------------------------------------------------------------------------------------
# the meta word's two counters (copterstep_internal.h): episode wraps, steps saturate
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("max_steps,sbits", [(1000, 11), (5, 4), (3000, 13)])
def test_episode_counter_is_a_full_32_bit_count_and_the_step_counter_saturates(max_steps, sbits):
    """ABI 5: the episode counter is a full 32-bit count -- its low 29 - S bits in the meta word, the rest in the
    tile's EPH row (ABI 4 wrapped it at 2^(29-S) - 1: the perturbation and random-action streams of an env repeated
    after that many episodes).  Envs parked just below 2^(29-S), just below 2^(30-S), just below 2^32 and at small
    numbers are flown across those boundaries under auto-reset: episode numbers, the Philox forces drawn for them and
    the `episodes started` statistic against the oracle's plain count.  The step counter still has S bits and
    saturates (upstream's and the oracle's never do: the documented cap is applied here, in the comparison)."""
    import torch
    from gpu_util import device_steps_cap, device_episode_bits
    n = 300
    env, orc = make_pair("lander3d", n, "float32", "next_step", seed=21, max_steps=max_steps)
    ebits = device_episode_bits(max_steps)
    assert ebits == 29 - sbits and device_steps_cap(max_steps) == (1 << sbits) - 1
    ep_mask = (1 << ebits) - 1
    env.reset()
    orc.reset()
    ep = np.full(n, ep_mask - 1, np.uint32)                # crosses into the EPH row within a few resets
    ep[::7] = ep_mask
    ep[1::7] = 5
    ep[2::7] = 2 * (ep_mask + 1) - 2                       # already has a high part; crosses the next multiple
    ep[3::7] = 0xFFFFFFFE                                  # the 32-bit wrap: ... 2^32 - 1, then 1
    ep[4::7] = 0x9E3779B9                                  # an arbitrary large number
    env.set_state(episode=ep)
    orc.episode[:] = ep
    # (the reset's perturbation is still pending: the device draws it where it is consumed, under the episode number
    # the env has THEN -- the oracle stores the force at reset time, so restate its draw for the new numbers)
    from oracle import refvec
    orc.force[:] = refvec.draw_forces(orc.seed, orc.env_ids, ep - np.uint32(1), orc.tp.initial_random_force).astype(orc.T)
    assert np.array_equal(env.get_state(only=("episode",))["episode"], ep)
    rng = np.random.default_rng(2)
    for t in range(60):
        a = rng.uniform(-1, 1, (n, 4)).astype(np.float32)
        got, want, _ = step_both(env, orc, a)
        assert_step_close(got, want, 2e-6, r_abs="auto", ctx="t=%d" % t)
        st = env.get_state()
        assert np.array_equal(st["episode"], orc.episode), t
        assert np.array_equal(st["force"].astype(np.float32), orc.force.astype(np.float32)), t
    e0, e1 = ep.astype(np.int64), orc.episode.astype(np.int64)
    assert np.any((e0 <= ep_mask) & (e1 > ep_mask)), "no env crossed 2^%d" % ebits
    assert np.any((e0 < 2 * (ep_mask + 1)) & (e0 > ep_mask) & (e1 >= 2 * (ep_mask + 1))), "no env crossed 2^%d" % (ebits + 1)
    assert np.any((e0 > 0xFFFFFF00) & (e1 < 100) & (e1 >= 1)), "no env wrapped from 2^32 - 1 to 1"
    assert_state_close(env, orc, MODE_TOL["float32"])
    assert float(to_np(env.batch_stats())[4]) == float(orc.episode.astype(np.float64).sum())
    # a checkpoint round trip keeps the full numbers (ABI 4 masked them silently)
    snap = env.get_state()
    env.set_state(**{k: v for k, v in snap.items()})
    assert np.array_equal(env.get_state(only=("episode",))["episode"], orc.episode)
    env.close()
    # saturation: nobody resets these envs
    env, orc = make_pair("hover3d", 64, "float32", "disabled", seed=2, max_steps=max_steps)
    env.reset()
    orc.reset()
    cap = device_steps_cap(max_steps)
    env.set_state(steps=np.full(64, cap - 2, np.int32))
    orc.steps[:] = cap - 2
    hover = np.full((64, 4), 0.0165, np.float32)
    for t in range(5):
        step_both(env, orc, hover)
    assert int(orc.steps.max()) == cap + 3                  # the oracle counts on, as upstream does (task.py:130)
    assert np.array_equal(env.get_state()["steps"], np.minimum(orc.steps, cap))
    with pytest.raises(Exception, match="steps out of range"):
        env.set_state(steps=np.full(64, cap + 1, np.int32))
    env.close()


def test_prev_shaping_travels_inside_the_r2_group():
    """prev_shaping is a word of the R2 group now: set / get round trip incl. NaN (= None), the reward of the next
    step is shaping - that value, in both word widths, and a Hover env keeps its NaN."""
    import torch
    for mode in ("float32", "float64"):
        env, orc = make_pair("lander3d", 130, mode, "disabled", seed=4)
        env.reset()
        orc.reset()
        prev = np.linspace(-300, -200, 130)
        prev[3] = np.nan
        env.set_state(prev_shaping=prev)
        orc.prev_shaping[:] = prev.astype(orc.T)
        got = env.get_state(only=("prev_shaping",))["prev_shaping"]
        assert np.array_equal(got, prev.astype(orc.T).astype(np.float64), equal_nan=True)
        a = np.full((130, 4), 0.0166, np.float32)
        g, w, _ = step_both(env, orc, a)
        assert_step_close(g, w, MODE_TOL[mode], ctx=mode)
        assert g[1][3] == 0.0                                      # prev_shaping None -> reward 0 (lander.py:58-61)
        assert_state_close(env, orc, MODE_TOL[mode])
        assert np.allclose(env.get_state()["prev_shaping"], orc.prev_shaping.astype(np.float64), rtol=1e-6)
        env.close()
    env, _ = make_pair("hover3d", 70, "float32", "next_step")
    env.reset()
    assert np.all(np.isnan(env.get_state()["prev_shaping"]))
    env.step(torch.zeros((70, 4), device=env.device))
    assert np.all(np.isnan(env.get_state()["prev_shaping"]))
    env.close()


@pytest.mark.parametrize("mode", ["float32", "float64"])
@pytest.mark.parametrize("substeps", [10, 3])
def test_an_envs_bytes_do_not_depend_on_its_wavefront_neighbours(substeps, mode):
    """With substeps > 1 a wavefront takes the free-flight call form when ALL its lanes qualify, the general form
    otherwise -- so WHICH form advances an env depends on its neighbours.  The two forms must leave the same bytes
    (sign of a zero included).  One batch against the same envs split at a point that is NOT a multiple of 64 (every
    env gets other neighbours; global ids keep the random draws): raw state words compared bit for bit."""
    import torch
    import gym_copter_amd
    n, cut = 4096 + 77, 1000 + 13
    kw = dict(task="lander3d", state_dtype=mode, seed=31, autoreset_mode="next_step", substeps=substeps)
    whole = gym_copter_amd.CopterVecEnv(num_envs=n, **kw)
    parts = [gym_copter_amd.CopterVecEnv(num_envs=cut, **kw),
             gym_copter_amd.CopterVecEnv(num_envs=n - cut, env_id_base=cut, **kw)]
    whole.reset()
    for p in parts:
        p.reset()
    rng = np.random.default_rng(9)
    hover = 0.016560178185018043
    # mostly level, quiet envs (zero roll / pitch torque: ax = ay = -0.0 at level attitude) with disturbed ones mixed
    # in at random places, so that wavefronts of both kinds exist and differ between the two groupings
    for t in range(120):
        a = np.full((n, 4), hover * (1 + 0.002 * np.sin(0.1 * t)), np.float32)
        wild = rng.random(n) < 0.03
        a[wild] = rng.uniform(-1, 1, (int(wild.sum()), 4)).astype(np.float32)
        tilt = rng.random(n) < 0.05
        a[tilt] *= np.array([1.0, 1.02, 1.02, 1.0], np.float32)
        at = torch.from_numpy(a).to(whole.device)
        ow = whole.step(at)
        op = [parts[0].step(at[:cut]), parts[1].step(at[cut:])]
        for k in range(4):
            joined = torch.cat([op[0][k], op[1][k]])
            wk = ow[k]
            if wk.dtype == torch.float32:            # bit patterns, not values: -0.0 != +0.0 here
                assert torch.equal(wk.view(torch.int32), joined.view(torch.int32)), (t, k)
            else:
                assert torch.equal(wk, joined), (t, k)
    sw = whole.get_state()
    sp = [p.get_state() for p in parts]
    for k in sw:
        joined = np.concatenate([sp[0][k], sp[1][k]], axis=-1)
        a64, b64 = np.ascontiguousarray(sw[k]), np.ascontiguousarray(joined)
        assert a64.tobytes() == b64.tobytes(), k
    # the batch did hold exact zeros of either sign somewhere (else the test shows nothing about them)
    x = sw["x"]
    assert np.any((x == 0) & np.signbit(x)) or np.any((x == 0) & ~np.signbit(x))
    whole.close()
    for p in parts:
        p.close()


# ---------------------------------------------------------------------------------------
# VERDICT round 3, weak #1b: where device and oracle "round differently once in 1e4 values"
# ---------------------------------------------------------------------------------------
def test_stored_word_codec_is_bit_exact_over_two_million_values():
    """The stored format itself (float32 word + 5 guard bits: encode on the device, decode on the device) is
    bit-identical to the oracle's model of it (refvec.guard_round) -- 2.1 M float64 values through cs_set_state /
    cs_get_state: random values over 60 decades, both signs, exact ties at the rounding position, values one
    float64 ulp either side of a tie, all-ones mantissas (carry into the exponent), zeros.  So a stored word that
    differs between device and oracle after a STEP is never the codec: it is the float64 value that went in."""
    import gym_copter_amd
    from oracle import refvec
    n = 175104                                      # 2736 tiles; 12 values per env
    rng = np.random.default_rng(77)
    v = rng.standard_normal((12, n)) * 10.0 ** rng.uniform(-30, 30, (12, n))
    bits = v.view(np.uint64).copy()
    k = n // 6
    tie = (bits[:, :k] & ~np.uint64(0xFFFFFF)) | np.uint64(0x800000)            # exactly half way
    bits[:, :k] = tie
    bits[:, k:2 * k] = tie + np.uint64(1)                                         # one ulp above a tie
    bits[:, 2 * k:3 * k] = tie - np.uint64(1)                                     # one ulp below
    bits[:, 3 * k:3 * k + 1000] |= np.uint64((1 << 52) - 1)                       # mantissa all ones: carry
    v = bits.view(np.float64).copy()
    v[:, 3 * k + 1000:3 * k + 1100] = 0.0
    v[:, 3 * k + 1100:3 * k + 1200] = -0.0
    want = refvec.guard_round(v)
    for task in ("lander3d", "hover3d"):
        env = gym_copter_amd.CopterVecEnv(task, n, state_dtype="float32")
        env.reset()
        env.set_state(x=v)
        got = env.get_state(only=("x",))["x"]
        assert got.view(np.uint64).tobytes() == want.view(np.uint64).tobytes(), task
        # and what an observation carries is the float32 rounding of exactly that value
        obs = to_np(env.state_tensors()["x"])
        with np.errstate(over="ignore"):
            assert np.array_equal(obs.view(np.uint32), want.astype(np.float32).view(np.uint32))
        env.close()
    env = gym_copter_amd.CopterVecEnv("lander3d", n, state_dtype="float32_rn")
    env.reset()
    env.set_state(x=v)
    got = env.get_state(only=("x",))["x"]
    with np.errstate(over="ignore"):
        assert np.array_equal(got.view(np.uint64), v.astype(np.float32).astype(np.float64).view(np.uint64))
    env.close()


def test_a_differing_stored_word_after_one_step_is_a_straddled_rounding_boundary():
    """Device and oracle start one step from IDENTICAL stored states (set through the bit-exact codec above) with the
    same actions.  Which operation makes their stored words differ "about once in 1e4 values" ?  Not
    the format (previous test) but the float64 value that is rounded into it: in the float32 state modes the device
    evaluates sin / cos with shorter polynomials (absolute error 1.4e-11 / 2.3e-13), which reaches
    the three translational velocities through the body-Z -> NED rotation; a value that close to a rounding boundary
    of the 29-bit format lands on the other side.  Asserted: positions and angles (x += dt * dx: one fused multiply-add
    of identical inputs) never differ; every differing word is within one unit of the format plus that 1e-10 of
    absolute slack; the rate per value is below 5e-3 (it is printed, per component); and in the float64 state mode
    (full fdlibm polynomials, no rounding step) the same step agrees to 1e-13."""
    import torch
    n = 131072
    rng = np.random.default_rng(5)
    x0 = np.zeros((12, n))
    x0[[0, 2]] = rng.uniform(-8, 8, (2, n))
    x0[4] = rng.uniform(-20, -1, n)
    x0[[1, 3, 5]] = rng.uniform(-3, 3, (3, n))
    x0[[6, 8]] = rng.uniform(-0.6, 0.6, (2, n))
    x0[10] = rng.uniform(-3, 3, n)
    x0[[7, 9, 11]] = rng.uniform(-1, 1, (3, n))
    a = rng.uniform(0.0, 0.05, (n, 4)).astype(np.float32)
    res = {}
    for mode in ("float32", "float64"):
        env, orc = make_pair("lander3d", n, mode, "disabled", seed=1)
        env.reset()
        orc.reset()
        env.set_state(x=x0, flags=np.zeros(n, np.uint8))
        start = env.get_state(only=("x",))["x"]
        orc.x[:] = start
        orc.pending[:] = False
        env.step(torch.from_numpy(a).to(env.device))
        orc.step(a.astype(np.float64))
        res[mode] = (env.get_state(only=("x",))["x"], orc.x.astype(np.float64).copy())
        env.close()
    got, want = res["float32"]
    diff = got.view(np.int64) != want.view(np.int64)
    per_slot = diff.mean(axis=1)
    print("stored words differing after one step, per component: " + " ".join("%.1e" % r for r in per_slot))
    print("overall: %d of %d (rate %.2e)" % (diff.sum(), diff.size, diff.mean()))
    assert not diff[[0, 2, 4, 6, 8, 10]].any()                  # positions and angles: identical inputs, one fma
    unit = 2.0 ** (np.floor(np.log2(np.maximum(np.abs(want), 1e-300))) - 28)      # one unit of the 29-bit format
    assert np.all(np.abs(got - want)[diff] <= unit[diff] + 1e-10)
    assert diff.mean() < 5e-3
    g64, w64 = res["float64"]
    assert np.max(np.abs(g64 - w64) / np.maximum(np.abs(w64), 1.0)) < 1e-13


def test_episode_counter_outgrows_the_meta_word_naturally_under_the_on_device_random_policy():
    """No parked counters: a step limit of 100 000 leaves the episode counter 11 bits in the meta word, and under
    the on-device random policy (episodes of ~7 steps) nearly every env passes episode 2 047 within 18 400 steps -- from
    there its number continues in the tile's EPH row (ABI 4 wrapped to 1 here).  cs_rollout_random in launches of 400
    steps against the oracle driven by the oracle's own draw of the same actions (keyed by seed, env id, EPISODE and
    step: a wrong episode number changes every action after it), every step's flags and the state after every launch;
    the random policy's and the reset perturbation's Philox counters both run through the boundary, inside the K-step
    kernel."""
    from oracle.refvec import draw_actions
    n, K, launches = 192, 400, 46
    env, orc = make_pair("lander3d", n, "float32", "next_step", seed=99, env_id_base=4096, max_steps=100000)
    from gpu_util import device_episode_bits
    assert device_episode_bits(100000) == 11
    env.reset()
    orc.reset()
    ids = np.arange(4096, 4096 + n)
    wrapped = np.zeros(n, bool)
    for launch in range(launches):
        obs_k, rew_k, term_k, trunc_k, act_k = (to_np(v) for v in env.rollout_random(K, return_actions=True))
        for k in range(K):
            before = orc.episode.copy()
            a = draw_actions(99, ids, orc.episode, orc.steps, 4)
            assert np.array_equal(a, act_k[k]), (launch, k)
            _, _, t, tr = orc.step(a.astype(np.float64))
            assert np.array_equal(term_k[k], t) and np.array_equal(trunc_k[k], tr), (launch, k)
            wrapped |= (before <= 2047) & (orc.episode > 2047)
        st = env.get_state()
        assert np.array_equal(st["episode"], orc.episode) and np.array_equal(st["steps"], orc.steps), launch
        assert_state_close(env, orc, 2e-6, ctx="launch %d" % launch)
    assert wrapped.mean() > 0.9 and orc.episode.max() > 2047 and orc.episode.min() >= 1
    env.close()
