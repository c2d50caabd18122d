"""Pins oracle/refcpu.py (scalar float64 restatement) bit-for-bit against golden
traces captured from the real reference by tests/golden/generate_golden.py."""
import numpy as np
import pytest

from conftest import load_cases
from oracle.refcpu import (AIRBORNE, CRASHED, DJI_PHANTOM, LANDED, RigidBody, TaskOracle,
                           TaskParams)

DYN = load_cases("dynamics_traces.npz")
ENV = load_cases("env_traces.npz", "variant_traces.npz")   # 3D tasks + 1D / 2D variants


@pytest.mark.parametrize("name", DYN.names())
def test_dynamics_trace_bit_exact(name):
    g = DYN[name]
    b = RigidBody(DJI_PHANTOM, int(g["fps"]))
    b.set_state(g["x0"])
    assert b.status == g["status0"]
    if np.any(g["force"]):
        b.perturb(g["force"])
    for t, m in enumerate(g["motors"]):
        b.set_motors(m)
        assert np.array_equal(b.x, g["x"][t]), (name, t)
        assert b.status == g["status"][t] and b.ticks == g["ticks"][t], (name, t)


@pytest.mark.parametrize("name", ENV.names())
def test_env_trace_bit_exact(name):
    g = ENV[name]
    f32_actions = bool(g["action_is_f32"])
    o = TaskOracle(str(g["task"]), TaskParams(initial_altitude=float(g["altitude"])),
                   action_dtype_passthrough=f32_actions)
    obs0 = o.reset(force_xyz=g["force"])
    assert obs0.dtype == np.float32 and np.array_equal(obs0, g["obs0"])
    assert o.steps == 1
    for t, a in enumerate(g["actions"]):
        if f32_actions:
            a = a.astype(np.float32)
        obs, r, done, trunc, info = o.step(a)
        assert trunc is False and info == {}
        assert np.array_equal(obs, g["obs"][t]), (name, t)
        assert r == g["reward"][t] and done == g["done"][t], (name, t)
        assert o.body.status == g["status"][t] and o.steps == g["steps"][t], (name, t)
        assert np.array_equal(o.body.x, g["x"][t]), (name, t)
        ps = g["prev_shaping"][t]
        assert (o.prev_shaping is None and np.isnan(ps)) or o.prev_shaping == ps


def test_known_answers():
    """Spot values observed on the real reference."""
    o = TaskOracle("lander3d")
    obs = o.reset(force_xyz=[0.0, 0.0, 0.0])
    assert np.array_equal(obs, np.array([0, 0, 0, 0, -10, 0, 0, 0, 0, 0], dtype=np.float32))
    assert o.prev_shaping == -250.0 and o.steps == 1 and o.body.status == AIRBORNE
    # forward Euler: first step from rest changes dz but not z
    o.step(1.625e-2 * np.ones(4))
    assert o.body.x[4] == -10.0 and abs(o.body.x[5] - 0.363923869e-2) < 1e-9
    # the reset perturbation is applied twice on the first integrated step
    o = TaskOracle("lander3d")
    o.reset(force_xyz=[2.0, 0.0, 0.0])
    o.step(np.zeros(4))
    assert o.body.x[1] == 2 * (2.0 / 1.380) * 0.01
    # free fall onto the ground crashes; status lags one step in `done`
    g = ENV["E02_lander_const"]
    fd = int(g["first_done"])
    assert g["status"][fd] == CRASHED and g["status"][fd - 1] == CRASHED and not g["done"][fd - 1]
    # a landed lander stays LANDED and keeps reporting done
    g = ENV["E07_lander_soft_landing"]
    assert g["status"][-1] == LANDED and g["done"][-1]
    # step limit fires on the 1000th user step
    assert ENV["E06_lander_hover_limit"]["first_done"] == 999


PID = load_cases("pid_traces.npz")


@pytest.mark.parametrize("name", PID.names())
def test_pid_heuristic_closed_loop_bit_exact(name):
    """oracle PidHeuristic + TaskOracle reproduce, bit for bit, the episode the reference's own
    controller classes (attic/mars/pidcontrollers) flew on the reference's live Lander."""
    from oracle.refcpu import PidGains, PidHeuristic
    g = PID[name]
    rk, pk = g["rate_gains"], g["pos_gains"]
    kw = dict(rate_kp=rk[0], rate_ki=rk[1], rate_kd=rk[2], pos_kp=pk[0], pos_ki=pk[1], pos_kd=pk[2], pos_target=pk[3])
    hover = "heuristic" in g and str(g["heuristic"]) == "hover"
    if hover:
        ak = g["alt_gains"]
        kw.update(heuristic="hover", alt_kp=ak[0], alt_ki=ak[1], alt_kd=ak[2], alt_target=ak[3])
    else:
        kw.update(descent_kp=g["descent_gains"][0], descent_kd=g["descent_gains"][1])
    gains = PidGains(**kw)
    pol = PidHeuristic(gains)
    env = TaskOracle("hover3d" if hover else "lander3d", TaskParams(initial_altitude=float(g["altitude"])))
    obs = env.reset(force_xyz=g["force"])
    for t in range(len(g["reward"])):
        a = pol.action(obs).astype(np.float32).astype(np.float64)    # the action space is float32
        assert np.array_equal(a, g["action"][t]), (name, t)
        obs, r, done, _, _ = env.step(a)
        assert np.array_equal(obs, g["obs"][t]) and r == g["reward"][t] and done == g["done"][t], (name, t)
        assert np.array_equal(env.body.x, g["x"][t]), (name, t)


VEH = load_cases("vehicle_traces.npz")


def _vehicle(g):
    from oracle.refcpu import VehicleParams
    v = g["vehicle"]
    return VehicleParams(*[float(t) for t in v[:9]]), float(v[9])


@pytest.mark.parametrize("name", VEH.names())
def test_other_vehicles_and_worlds_bit_exact(name):
    """Episodes the reference's Lander flew with other vehicle_params dicts / gravity constants."""
    g = VEH[name]
    vp, grav = _vehicle(g)
    o = TaskOracle("lander3d", TaskParams(initial_altitude=float(g["altitude"])), vp=vp, g=grav)
    assert np.array_equal(o.reset(force_xyz=g["force"]), g["obs0"])
    for t, a in enumerate(g["actions"]):
        obs, r, done, _, _ = o.step(a)
        assert np.array_equal(obs, g["obs"][t]) and r == g["reward"][t] and done == g["done"][t], (name, t)
        assert o.body.status == g["status"][t] and np.array_equal(o.body.x, g["x"][t]), (name, t)


POSE = load_cases("pose_traces.npz")


@pytest.mark.parametrize("name", POSE.names())
def test_pose_reset_bit_exact(name):
    """_Task._reset(pose=..., perturb=...) then ordinary steps: scalar oracle vs the reference."""
    g = POSE[name]
    o = TaskOracle(str(g["task"]))
    obs0 = o.reset(force_xyz=g["force"], pose=g["pose"], perturb=bool(g["perturb"]))
    assert np.array_equal(obs0, g["obs0"]) and np.array_equal(o.body.x, g["x0"])
    for t, a in enumerate(g["actions"]):
        obs, r, done, _, _ = o.step(a)
        assert np.array_equal(obs, g["obs"][t]) and r == g["reward"][t] and done == g["done"][t], (name, t)
        assert o.body.status == g["status"][t] and o.steps == g["steps"][t], (name, t)
        assert np.array_equal(o.body.x, g["x"][t]), (name, t)


MARS = load_cases("mars_traces.npz")


def mars_body(g):
    """RigidBody configured as the retired Mars model from a golden case's `vehicle` row
    (B, D, M, L, Ix, Iy, Iz, Jr, maxrpm, G, rho, C_L)."""
    from oracle.refcpu import VehicleParams
    v = g["vehicle"]
    return RigidBody(VehicleParams(*[float(x) for x in v[:9]]), int(g["fps"]), g=float(v[9]),
                     mars=(float(v[10]), float(v[11])))


@pytest.mark.parametrize("name", MARS.names())
def test_mars_dynamics_trace_bit_exact(name):
    """The lift-coefficient thrust law, air density and the live rotor-inertia term of
    attic/mars/dynamics (setMotors + update per tick), bit for bit."""
    g = MARS[name]
    b = mars_body(g)
    b.set_state(g["x0"])
    assert b.status == g["status0"]
    if np.any(g["force"]):
        b.perturb(g["force"])
    for t, m in enumerate(g["motors"]):
        b.set_motors(m)
        assert np.array_equal(b.x, g["x"][t]), (name, t)
        assert b.status == g["status"][t], (name, t)
