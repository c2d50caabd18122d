"""oracle/refvec.py (batch checker used on the GPU box) vs the golden traces and vs
the scalar oracle; plus the batch-only features it models (Philox draw, auto-reset,
float32 state words)."""
import numpy as np
import pytest

from conftest import load_cases
from oracle import refvec
from oracle.refcpu import AIRBORNE, TaskOracle, TaskParams
from oracle.refvec import VecOracle, draw_forces, philox2x32_10

DYN = load_cases("dynamics_traces.npz")
ENV = load_cases("env_traces.npz", "variant_traces.npz")   # 3D tasks + 1D / 2D variants


def test_philox_known_answers():
    """Random123 kat_vectors for philox2x32-10."""
    kat = [((0, 0), 0, (0xff1dae59, 0x6cd10df2)),
           ((0xffffffff, 0xffffffff), 0xffffffff, (0x2c3f628b, 0xab4fd7ad)),
           ((0x243f6a88, 0x85a308d3), 0x13198a2e, (0xdd7ce038, 0xf62a4c12))]
    for ctr, key, want in kat:
        got = philox2x32_10(*ctr, key)
        assert tuple(int(v) for v in got) == want


def test_draw_forces_distribution_and_keys():
    f = draw_forces(1234, np.arange(200000), 5, 30)
    assert f.shape == (3, 200000) and f.min() >= -30 and f.max() < 30
    assert abs(f.mean()) < 0.1 and abs(f.std() - 60 / np.sqrt(12)) < 0.1
    # 21-bit grid over 60 N: exact in float64 whether or not the multiply-add is fused
    assert np.array_equal(f * 2.0 ** 21 / 4, np.round(f * 2.0 ** 21 / 4))
    # keyed by global env id, episode number and seed
    g = draw_forces(1234, np.arange(100, 200), 5, 30)
    assert np.array_equal(g, f[:, 100:200])
    assert not np.array_equal(draw_forces(1234, np.arange(100), 6, 30), f[:, :100])
    assert not np.array_equal(draw_forces(1235, np.arange(100), 5, 30), f[:, :100])
    # every bit of the 64-bit seed matters (the key is a half of splitmix64(seed), not lo ^ hi)
    assert not np.array_equal(draw_forces(0x0000000100000001, np.arange(100), 5, 30), draw_forces(0, np.arange(100), 5, 30))
    assert not np.array_equal(draw_forces(0x0000000200000001, np.arange(100), 5, 30),
                              draw_forces(0x0000000100000002, np.arange(100), 5, 30))
    e = np.arange(100) % 7
    h = draw_forces(1234, np.arange(100), e, 30)
    for i in range(100):
        assert np.array_equal(h[:, i], draw_forces(1234, [i], int(e[i]), 30)[:, 0])


def _env_groups():
    groups = {}
    for c in ENV.names():
        g = ENV[c]
        if bool(g["action_is_f32"]):
            continue
        groups.setdefault((str(g["task"]), float(g["altitude"])), []).append(c)
    return groups


@pytest.mark.parametrize("key", sorted(_env_groups()))
def test_vec_env_traces_bit_exact_in_batch(key):
    """All golden episodes of one task stepped as ONE batch reproduce the reference
    bit-for-bit (float64 state words, auto-reset disabled)."""
    task, alt = key
    cs = _env_groups()[key]
    n = len(cs)
    T = max(len(ENV[c]["reward"]) for c in cs)
    acts = np.zeros((T, n, ENV[cs[0]]["actions"].shape[1]))
    forces = np.zeros((3, n))
    for i, c in enumerate(cs):
        a = ENV[c]["actions"]
        acts[:len(a), i] = a
        forces[:, i] = ENV[c]["force"]
    o = VecOracle(task, n, TaskParams(initial_altitude=alt))
    obs0 = o.reset(forces=forces)
    for i, c in enumerate(cs):
        assert np.array_equal(obs0[i], ENV[c]["obs0"])
    for t in range(T):
        obs, r, term, trunc = o.step(acts[t])
        assert not trunc.any()
        for i, c in enumerate(cs):
            g = ENV[c]
            if t < len(g["reward"]):
                assert np.array_equal(obs[i], g["obs"][t]), (c, t)
                assert r[i] == g["reward"][t] and term[i] == g["done"][t], (c, t)
                assert o.status[i] == g["status"][t] and o.steps[i] == g["steps"][t], (c, t)
                assert np.array_equal(o.x[:, i], g["x"][t]), (c, t)


@pytest.mark.parametrize("fps", [100, 1000])
def test_vec_dynamics_traces_bit_exact_in_batch(fps):
    cs = [c for c in DYN.names() if int(DYN[c]["fps"]) == fps]
    n = len(cs)
    T = max(len(DYN[c]["status"]) for c in cs)
    o = VecOracle("lander3d", n, TaskParams(frames_per_second=fps))
    motors = np.zeros((T, n, 4))
    for i, c in enumerate(cs):
        g = DYN[c]
        o.x[:, i] = g["x0"]
        o.status[i] = g["status0"]
        o.force[:, i] = g["force"][:3]
        o.pending[i] = bool(np.any(g["force"]))
        motors[:len(g["motors"]), i] = g["motors"]
    for t in range(T):
        o.set_motors(motors[t])
        for i, c in enumerate(cs):
            g = DYN[c]
            if t < len(g["status"]):
                assert np.array_equal(o.x[:, i], g["x"][t]), (c, t)
                assert o.status[i] == g["status"][t], (c, t)
                assert o.ticks[i] == g["ticks"][t], (c, t)      # Dynamics._ticks: a contact freeze does not tick


def test_substeps_match_fps1000_trace():
    """10 inner substeps at dt=1e-3 == Dynamics(params, 1000) called 10x per env step."""
    g = DYN["D10_fps1000"]
    o = VecOracle("lander3d", 1, substeps=10)
    o.reset(forces=g["force"][:3, None])
    assert np.array_equal(o.x[:, 0], g["x0"])
    for s in range(1000):
        o.step(np.tile(g["motors"][10 * s], (1, 1)))
        assert np.array_equal(o.x[:, 0], g["x"][10 * s + 9]), s


def _worst_scaled_error(case, mode, seed=0):
    g = ENV[case]
    o = VecOracle("lander3d", 1, store_mode=mode, seed=seed)
    o.reset(forces=g["force"][:, None])
    worst = 0.0
    for t in range(int(g["first_done"])):
        o.step(g["actions"][t][None])
        ref = g["x"][t]
        err = np.abs(o.x[:, 0].astype(np.float64) - ref) / np.maximum(np.abs(ref), 1.0)
        worst = max(worst, err.max())
    return worst


@pytest.mark.parametrize("case", ["E06_lander_hover_limit", "E09_lander_noisy_hover",
                                  "E02_lander_const"])
def test_float32_words_with_guard_bits_track_reference(case):
    """float64 arithmetic + float32 state words with 5 guard bits (the default device
    format) stay within 2e-6 * max(|ref|, 1) of the float64 reference over a whole
    (up to 1000-step) episode -- five times inside the 1e-5 parity bar."""
    assert _worst_scaled_error(case, "float32") < 2e-6


def test_plain_float32_words_miss_the_bar_on_constant_thrust():
    """Why guard bits are the default: with bare float32 words (round-to-nearest) 1000 accumulations x += dt*dxdt lose ~10-500 ulp(x): the constant-thrust
    episodes drift to >= 1e-5 while noisy ones stay inside."""
    assert _worst_scaled_error("E06_lander_hover_limit", "float32_rn") > 1e-5
    assert _worst_scaled_error("E02_lander_const", "float32_rn") > 1e-5
    assert _worst_scaled_error("E09_lander_noisy_hover", "float32_rn") < 1e-5


def _scalar_rollout(task, forces_per_episode, actions, tp=TaskParams()):
    """Scalar oracle with manual reset after done: NEXT_STEP convention."""
    o = TaskOracle(task, tp)
    ep = 0
    obs = o.reset(force_xyz=forces_per_episode[ep])
    out = []
    need_reset = False
    for a in actions:
        if need_reset:
            ep += 1
            obs = o.reset(force_xyz=forces_per_episode[ep])
            out.append((obs, 0.0, False))
            need_reset = False
            continue
        obs, r, done, _, _ = o.step(a)
        out.append((obs, r, done))
        need_reset = done
    return out


@pytest.mark.parametrize("task", ["lander3d", "hover3d"])
def test_autoreset_next_step_matches_scalar_oracle(task):
    rng = np.random.default_rng(5)
    n, T, seed = 6, 60, 99
    acts = rng.uniform(-1, 1, (T, n, 4)).astype(np.float32).astype(np.float64)
    v = VecOracle(task, n, autoreset=refvec.AUTORESET_NEXT_STEP, seed=seed, env_id_base=1000)
    obs0 = v.reset()
    # the forces the batch oracle will draw are a pure function of (seed, env id, episode #)
    got = []
    for t in range(T):
        got.append(v.step(acts[t]))
    for i in range(n):
        forces = [draw_forces(seed, [1000 + i], e, 30)[:, 0] for e in range(T)]
        want = _scalar_rollout(task, forces, acts[:, i])
        for t in range(T):
            obs, r, term, trunc = got[t]
            assert np.array_equal(obs[i], want[t][0]), (i, t)
            assert r[i] == want[t][1] and term[i] == want[t][2], (i, t)
    assert sum(int(g[2].sum()) for g in got) > n      # several episodes ended


def test_autoreset_same_step_and_truncation():
    rng = np.random.default_rng(6)
    n, T = 4, 40
    acts = rng.uniform(-1, 1, (T, n, 4))
    v = VecOracle("lander3d", n, autoreset=refvec.AUTORESET_SAME_STEP, seed=3)
    v.reset()
    ends = 0
    for t in range(T):
        obs, r, term, trunc = v.step(acts[t])
        for i in np.flatnonzero(term):
            # returned obs is already the reset observation; state is fresh
            assert np.array_equal(obs[i], np.array([0, 0, 0, 0, -10, 0, 0, 0, 0, 0], np.float32))
            assert v.steps[i] == 1 and v.status[i] == AIRBORNE and v.pending[i]
            ends += 1
    assert ends > 0
    # time limit reported as truncation when asked for
    import os
    hov = float(np.load(os.path.join(os.path.dirname(__file__), "golden", "meta.npz"))["hover_motor"])
    v = VecOracle("hover3d", 1, TaskParams(max_steps=20), time_limit_truncates=True)
    v.reset(forces=np.zeros((3, 1)))
    for t in range(20):
        obs, r, term, trunc = v.step(np.full((1, 4), hov))
        assert trunc[0] == (t == 19) and not term[0]


def test_vecpid_matches_scalar_heuristic_closed_loop():
    """VecPid + VecOracle(float64) == PidHeuristic + TaskOracle lane by lane, bit for bit."""
    from oracle.refcpu import PidGains, PidHeuristic
    from oracle.refvec import VecPid
    n, T = 5, 160
    for gains in (PidGains(), PidGains(rate_kp=0.002, rate_kd=0.002, pos_kp=0.0002, pos_ki=0.0, pos_kd=0.0,
                                       descent_kp=0.004, descent_kd=0.012),
                  PidGains(rate_ki=0.05, pos_ki=0.3),
                  PidGains(heuristic="hover"),
                  PidGains(heuristic="hover", alt_kp=0.02, alt_ki=5.0, rate_kp=0.002, rate_kd=0.002, rate_ki=0.01,
                           pos_kp=0.0002, pos_ki=0.0, pos_kd=0.0)):
        task = "hover3d" if gains.heuristic == "hover" else "lander3d"
        vec = VecOracle(task, n, store_mode="float64", seed=11)
        obs = vec.reset()
        forces = vec.force[:3].astype(np.float64).T.copy()
        pid = VecPid(n, gains)
        scal = []
        for i in range(n):
            env = TaskOracle(task)
            o = env.reset(force_xyz=forces[i])
            scal.append((env, PidHeuristic(gains), o))
        for t in range(T):
            a = pid.action(obs)
            obs, r, term, _ = vec.step(a)
            for i, (env, pol, o) in enumerate(scal):
                ai = pol.action(o).astype(np.float32)
                assert np.array_equal(ai, a[i]), (t, i)
                o2, ri, di, _, _ = env.step(ai.astype(np.float64))
                assert np.array_equal(o2, obs[i]) and ri == r[i] and di == term[i], (t, i)
                scal[i] = (env, pol, o2)


def test_vec_per_env_vehicles_bit_exact_in_batch():
    """All golden other-vehicle / other-world episodes as ONE batch with per-env parameter arrays."""
    from conftest import load_cases
    from oracle.refcpu import VehicleParams
    VEH = load_cases("vehicle_traces.npz")
    for alt in (10.0, 0.05):
        cs = [c for c in VEH.names() if float(VEH[c]["altitude"]) == alt]
        n = len(cs)
        veh = np.stack([VEH[c]["vehicle"] for c in cs], axis=1)          # [10, n]
        vp = VehicleParams(*[veh[j].copy() for j in range(9)])
        T = max(len(VEH[c]["reward"]) for c in cs)
        acts = np.zeros((T, n, 4))
        forces = np.zeros((3, n))
        for i, c in enumerate(cs):
            acts[:len(VEH[c]["actions"]), i] = VEH[c]["actions"]
            forces[:, i] = VEH[c]["force"]
        o = VecOracle("lander3d", n, TaskParams(initial_altitude=alt), vp=vp, g=veh[9].copy())
        o.reset(forces=forces)
        for t in range(T):
            with np.errstate(all="ignore"):      # short episodes keep free-running (and diverge) past their end
                obs, r, term, _ = o.step(acts[t])
            for i, c in enumerate(cs):
                g = VEH[c]
                if t < len(g["reward"]):
                    assert np.array_equal(obs[i], g["obs"][t]) and r[i] == g["reward"][t], (c, t)
                    assert term[i] == g["done"][t] and np.array_equal(o.x[:, i], g["x"][t]), (c, t)


KNOWN_ACTION = [0.721649169921875, -0.21759033203125, -0.518310546875, -0.666656494140625]


def test_draw_actions_spec():
    """The on-device random policy's draw: on the 2^-15 grid in [-1, 1), a pure function of (seed,
    env id, episode, step), distinct from the reset-force stream, roughly uniform."""
    from oracle.refvec import draw_actions
    ids = np.arange(1000, 1000 + 4096, dtype=np.uint64)
    a = draw_actions(7, ids, 3, 11)
    assert a.shape == (4096, 4) and a.dtype == np.float32
    assert a.min() >= -1 and a.max() < 1 and np.array_equal(a * 32768, np.round(a * 32768))
    assert np.array_equal(a[100:200], draw_actions(7, ids[100:200], 3, 11))           # position invariant
    assert not np.array_equal(a, draw_actions(7, ids, 3, 12)) and not np.array_equal(a, draw_actions(7, ids, 4, 11))
    assert not np.array_equal(a, draw_actions(8, ids, 3, 11))
    assert np.array_equal(draw_actions(7, ids, 3, 11, act_dim=2), a[:, :2])
    assert abs(a.mean()) < 0.02 and abs(a.std() - 1 / np.sqrt(3)) < 0.01
    # known answer (first env, seed 1234, episode 1, step 1), pinned when the spec was written
    k = draw_actions(1234, [0], 1, 1)[0]
    assert np.array_equal(k, np.float32(KNOWN_ACTION)), k.tolist()


def test_vec_pose_resets_bit_exact_in_batch():
    """All pose-reset golden episodes of a task as one batch (per-env poses; with / without perturbation)."""
    from conftest import load_cases
    POSE = load_cases("pose_traces.npz")
    for task in ("lander3d", "hover3d"):
        for perturb in (True, False):
            cs = [c for c in POSE.names() if str(POSE[c]["task"]) == task and bool(POSE[c]["perturb"]) == perturb]
            if not cs:
                continue
            n = len(cs)
            poses = np.stack([POSE[c]["pose"] for c in cs], axis=1)
            forces = np.stack([POSE[c]["force"] for c in cs], axis=1)
            o = VecOracle(task, n)
            obs0 = o.reset(forces=forces, poses=poses, perturb=perturb)
            T = max(len(POSE[c]["reward"]) for c in cs)
            acts = np.zeros((T, n, 4))
            for i, c in enumerate(cs):
                assert np.array_equal(obs0[i], POSE[c]["obs0"]), c
                acts[:len(POSE[c]["actions"]), i] = POSE[c]["actions"]
            for t in range(T):
                with np.errstate(all="ignore"):
                    obs, r, term, _ = o.step(acts[t])
                for i, c in enumerate(cs):
                    g = POSE[c]
                    if t < len(g["reward"]):
                        assert np.array_equal(obs[i], g["obs"][t]) and r[i] == g["reward"][t], (c, t)
                        assert term[i] == g["done"][t] and o.status[i] == g["status"][t], (c, t)
                        assert np.array_equal(o.x[:, i], g["x"][t]), (c, t)


def test_vec_mars_dynamics_bit_exact_in_batch():
    """All Mars-model golden traces (lift-coefficient thrust law, per-env air density / gravity,
    live rotor-inertia term) as ONE batch with per-env parameter arrays."""
    from conftest import load_cases
    from oracle.refcpu import VehicleParams
    M = load_cases("mars_traces.npz")
    cs = M.names()
    n = len(cs)
    veh = np.stack([M[c]["vehicle"] for c in cs], axis=1)            # [12, n]
    o = VecOracle("lander3d", n, vp=VehicleParams(*[veh[j].copy() for j in range(9)]), g=veh[9].copy(),
                  mars=(veh[10].copy(), veh[11].copy()))
    o.x[:] = np.stack([M[c]["x0"] for c in cs], axis=1)
    o.status[:] = [int(M[c]["status0"]) for c in cs]
    o.force[:] = np.stack([M[c]["force"][:3] for c in cs], axis=1)
    o.pending[:] = [bool(np.any(M[c]["force"])) for c in cs]
    T = max(len(M[c]["motors"]) for c in cs)
    for t in range(T):
        m = np.stack([M[c]["motors"][min(t, len(M[c]["motors"]) - 1)] for c in cs])
        o.set_motors(m)
        for i, c in enumerate(cs):
            g = M[c]
            if t < len(g["status"]):
                assert np.array_equal(o.x[:, i], g["x"][t]), (c, t)
                assert o.status[i] == g["status"][t], (c, t)
