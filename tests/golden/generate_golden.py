#!/usr/bin/env python3
"""
DEV-ONLY fixture generator.  Runs ONLY in the build container, where the
upstream reference checkout is mounted read-only at /root/reference.  It
imports the reference's own NumPy implementation, drives it with recorded
inputs and writes small golden .npz files next to this script.  Nothing in
here travels as code to the GPU box except this script itself (as provenance);
the GPU box only ever reads the .npz data.

What is imported from the reference (never copied):
  * gym_copter/dynamics/__init__.py            -> class Dynamics   (D-series)
  * gym_copter/dynamics/vehicles/dji_phantom.py -> vehicle_params
  * gym_copter/envs/task.py, envs/lander.py    -> _Task, Lander    (E-series)
`gymnasium` is not installed in the container, and the reference only uses a
handful of names from it (Env, spaces.Box, utils.EzPickle, utils.seeding,
envs.registration.register), so a tiny in-memory stand-in for those names is
registered in sys.modules before the import.  The stand-in holds no reference
code.

Hover3D (attic/gym_copter/envs/hover3d.py + hover.py) cannot be imported (it
needs modules that no longer exist upstream), so its fixture is produced by a
subclass of the *imported live* _Task whose two overrides restate
hover.py:18-21 (reward == 1) and hover3d.py:32-37 (12-component observation).

Further series, all flown on the imported live classes:
  * P (pid_traces.npz): closed loop under the reference's own PID controller
    classes, loaded from attic/mars/pidcontrollers by file path and wired as
    attic/mars/lander3d.py:64-87 does;
  * V (variant_traces.npz): the 1D / 2D variants -- subclasses of the live Lander
    whose _get_motors / _get_state hooks restate attic lander1d.py:43-48,
    lander2d.py:43-50, hover1d.py:44-50, hover2d.py:44-50;
  * W (vehicle_traces.npz): the live Lander with other `vehicle_params` dicts and
    gravity constants (the module global of envs/task.py and Dynamics.G are
    swapped for the duration of the run);
  * M (mars_traces.npz): the retired Mars dynamics, attic/mars/dynamics/__init__.py +
    ingenuity.py loaded by file path (NumPy only): the lift-coefficient thrust law
    Lift = 0.5*rho*S*C_L*(omega*L/2)^2 with world parameters G / rho, and the
    rotor-inertia term with Omega = u4(omegas); setMotors() + update() per tick.

Inputs are made float32-representable (actions, perturbation forces, vehicle
parameters) so that the fp32 device path can be fed bit-identical inputs; the
reference still computes everything in float64.

Usage:  python tests/golden/generate_golden.py   (rewrites tests/golden/*.npz)
"""

import importlib
import importlib.util
import os
import sys
import types

import numpy as np

REF = os.environ.get("COPTER_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))


# --------------------------------------------------------------------------
# in-memory gymnasium stand-in (only the names the reference touches)
# --------------------------------------------------------------------------
def _install_gymnasium_standin():
    if "gymnasium" in sys.modules:
        return
    gym = types.ModuleType("gymnasium")

    class Env:
        @property
        def unwrapped(self):
            return self

        def close(self):
            pass

    class Box:
        def __init__(self, low, high, shape=None, dtype=np.float32):
            self.low, self.high, self.shape, self.dtype = low, high, shape, dtype

    class EzPickle:
        def __init__(self, *a, **k):
            pass

    spaces = types.ModuleType("gymnasium.spaces")
    spaces.Box = Box
    utils = types.ModuleType("gymnasium.utils")
    utils.EzPickle = EzPickle
    seeding = types.ModuleType("gymnasium.utils.seeding")
    seeding.np_random = lambda seed=None: (np.random.default_rng(seed), seed)
    utils.seeding = seeding
    envs = types.ModuleType("gymnasium.envs")
    registration = types.ModuleType("gymnasium.envs.registration")
    registration.register = lambda **kw: None
    envs.registration = registration
    gym.Env, gym.spaces, gym.utils, gym.envs = Env, spaces, utils, envs
    sys.modules.update({
        "gymnasium": gym, "gymnasium.spaces": spaces, "gymnasium.utils": utils,
        "gymnasium.utils.seeding": seeding, "gymnasium.envs": envs,
        "gymnasium.envs.registration": registration})


def load_reference():
    _install_gymnasium_standin()
    if REF not in sys.path:
        sys.path.insert(0, REF)
    dyn = importlib.import_module("gym_copter.dynamics")
    veh = importlib.import_module("gym_copter.dynamics.vehicles.dji_phantom")
    task = importlib.import_module("gym_copter.envs.task")
    lander = importlib.import_module("gym_copter.envs.lander")
    return dyn.Dynamics, veh.vehicle_params, task._Task, lander.Lander


def f32r(a):
    """Round to float32 and return as float64 (exactly representable both ways)."""
    return np.asarray(a, dtype=np.float32).astype(np.float64)


# hover motor value: B*4*w^2 = M*G  ->  m = sqrt(M*G/(4B)) / (maxrpm*pi/30)
def hover_motor(vp, G=9.80665):
    return float(np.sqrt(vp["M"] * G / (4 * vp["B"])) / (vp["maxrpm"] * np.pi / 30))


# --------------------------------------------------------------------------
# D-series: Dynamics.setMotors() traces
# --------------------------------------------------------------------------
def run_dynamics(Dynamics, vp, fps, x0, force, motors):
    """motors: [T,4] float64.  Returns per-call x[T,12], status[T], ticks[T]."""
    d = Dynamics(vp, fps)
    d.setState(np.array(x0, dtype=np.float64))
    status0 = d.getStatus()
    if force is not None:
        d.perturb(np.array(force, dtype=np.float64))
    T = motors.shape[0]
    xs = np.zeros((T, 12))
    st = np.zeros(T, dtype=np.int8)
    tk = np.zeros(T, dtype=np.int64)
    for t in range(T):
        d.setMotors(motors[t])
        xs[t] = d._x
        st[t] = d.getStatus()
        tk[t] = d._ticks
    return dict(fps=np.int64(fps), x0=np.array(x0, dtype=np.float64),
                status0=np.int8(status0),
                force=np.zeros(6) if force is None else np.array(force, dtype=np.float64),
                motors=motors, x=xs, status=st, ticks=tk)


def d_series(Dynamics, vp):
    rng = np.random.default_rng(20240117)
    hov = hover_motor(vp)
    cases = {}

    def airborne(alt=10.0, **kw):
        x = np.zeros(12)
        x[4] = -alt
        for k, v in kw.items():
            x[int(k[1:])] = v
        return x

    T = 1000
    ones = np.ones((T, 4))
    # D01 constant sub-hover thrust (lander.py's MOTORVAL), xyz perturbation
    cases["D01_const_motorval"] = run_dynamics(
        Dynamics, vp, 100, airborne(), f32r([12.5, -7.25, 3.0, 0, 0, 0]),
        f32r(1.625e-2) * ones)
    # D02 near hover for 1000 calls (netz = catastrophic cancellation case)
    cases["D02_near_hover"] = run_dynamics(
        Dynamics, vp, 100, airborne(), f32r([-3.5, 21.0, -29.75, 0, 0, 0]),
        f32r(hov) * ones)
    # D03 slightly asymmetric motors: roll/pitch/yaw torques all non-zero
    m = f32r(hov * np.array([1.0, 1.002, 0.999, 1.001]))
    cases["D03_asymmetric"] = run_dynamics(
        Dynamics, vp, 100, airborne(50.0), f32r([5, 5, 5, 0, 0, 0]), m * ones)
    # D04 random motor sequence around hover with non-zero initial rates
    mseq = f32r(hov * (1.0 + 0.02 * rng.standard_normal((T, 4))))
    cases["D04_random_motors"] = run_dynamics(
        Dynamics, vp, 100,
        airborne(200.0, i1=0.5, i3=-0.25, i6=0.05, i7=0.01, i8=-0.03, i9=0.02, i10=0.3, i11=-0.1),
        f32r([29.5, -29.5, 10.0, 0, 0, 0]), mseq)
    # D05 take-off from LANDED (z = 0): motors ramp from 0 to above hover
    ramp = f32r(np.linspace(0.0, 1.5 * hov, 300))[:, None] * np.ones((300, 4))
    cases["D05_takeoff"] = run_dynamics(Dynamics, vp, 100, np.zeros(12), None, ramp)
    # D06 ground contact -> CRASHED (free fall from 3 m, motors off)
    cases["D06_crash"] = run_dynamics(
        Dynamics, vp, 100, airborne(3.0), None, np.zeros((200, 4)))
    # D07 ground contact -> LEVELING -> LANDED (0.05 m, slightly below hover)
    cases["D07_soft_landing"] = run_dynamics(
        Dynamics, vp, 100, airborne(0.05, i6=0.01), None, f32r(1.6e-2) * np.ones((300, 4)))
    # D08 crash by lateral speed / by roll angle at contact
    cases["D08_crash_lateral"] = run_dynamics(
        Dynamics, vp, 100, airborne(0.05, i3=2.5), None, f32r(1.6e-2) * np.ones((200, 4)))
    cases["D09_crash_roll"] = run_dynamics(
        Dynamics, vp, 100, airborne(0.05, i6=0.9), None, f32r(1.6e-2) * np.ones((200, 4)))
    # D10 fps = 1000 (oracle for the 10-substep configuration), 10 000 calls
    cases["D10_fps1000"] = run_dynamics(
        Dynamics, vp, 1000, airborne(), f32r([12.5, -7.25, 3.0, 0, 0, 0]),
        f32r(hov * 0.999) * np.ones((10000, 4)))
    # D11 large angles (wrap well past pi) -- exercises sin/cos range reduction
    cases["D11_spin"] = run_dynamics(
        Dynamics, vp, 100,
        airborne(5000.0, i6=2.5, i8=-1.0, i10=40.0, i11=25.0),
        None, f32r(hov) * ones)
    # D12 full-range motors (0..1): huge thrust, values ~1e5
    cases["D12_full_range"] = run_dynamics(
        Dynamics, vp, 100, airborne(), f32r([1, 2, 3, 0, 0, 0]),
        f32r(rng.uniform(0, 1, (200, 4))))
    return cases


# --------------------------------------------------------------------------
# E-series: env-level traces
# --------------------------------------------------------------------------
def run_env(env, actions, seed, altitude=None, extra_after_done=25, action_dtype=np.float64, pose=None,
            perturb=True):
    """Drive env.reset() + env.step() with the recorded actions.

    The drawn perturbation force is rounded to float32 and re-installed before
    the first step so that an fp32 device path can be given identical inputs.
    Stepping continues `extra_after_done` steps past the first done (the
    reference allows that; it is what autoreset=DISABLED must reproduce)."""
    if altitude is not None:
        env.set_altitude(altitude)
    np.random.seed(seed)
    if pose is None:
        obs0, _ = env.reset()
    else:      # _Task._reset(pose=(x, y, altitude, phi_deg, theta_deg), perturb=...), as lander.py:85 uses it
        obs0, _ = env._reset(pose=tuple(float(v) for v in pose), perturb=perturb)
    d = env.dynamics
    force = f32r(d._perturb * d.M)
    d.perturb(force.copy())
    rec = dict(obs=[], reward=[], done=[], status=[], steps=[], prev_shaping=[], x=[])
    x0 = d._x.copy()
    T = actions.shape[0]
    first_done = -1
    t = 0
    while t < T:
        a = actions[t].astype(action_dtype)
        obs, reward, done, trunc, info = env.step(a)
        assert trunc is False and info == {}
        rec["obs"].append(obs)
        rec["reward"].append(float(reward))
        rec["done"].append(bool(done))
        rec["status"].append(int(d.getStatus()))
        rec["steps"].append(int(env.steps))
        ps = getattr(env, "prev_shaping", None)
        rec["prev_shaping"].append(np.nan if ps is None else float(ps))
        rec["x"].append(d._x.copy())
        if done and first_done < 0:
            first_done = t
        t += 1
        if first_done >= 0 and t > first_done + extra_after_done:
            break
    n = len(rec["reward"])
    return dict(
        task=np.array(getattr(env, "TASK_NAME", "hover3d" if len(obs0) == 12 else "lander3d")),
        seed=np.int64(seed), altitude=np.float64(env.initial_altitude),
        force=force[:3], actions=actions[:n].astype(np.float64),
        action_is_f32=np.bool_(action_dtype == np.float32),
        obs0=np.asarray(obs0, dtype=np.float32), first_done=np.int64(first_done),
        pose=np.asarray(pose if pose is not None else [0, 0, env.initial_altitude, 0, 0], dtype=np.float64),
        perturb=np.bool_(perturb), x0=np.asarray(x0),
        obs=np.asarray(rec["obs"], dtype=np.float32), reward=np.asarray(rec["reward"]),
        done=np.asarray(rec["done"]), status=np.asarray(rec["status"], dtype=np.int8),
        steps=np.asarray(rec["steps"], dtype=np.int64),
        prev_shaping=np.asarray(rec["prev_shaping"]), x=np.asarray(rec["x"]))


def make_hover_ref(_Task):
    class HoverRef(_Task):
        # restates attic hover.py:18-21 and hover3d.py:32-37 on the live _Task
        def __init__(self):
            _Task.__init__(self, 12, 4)

        def reset(self, seed=None, options=None):
            return _Task._reset(self, seed, options)

        def _get_reward(self, status, state, d, x, y):
            return 1

        def _get_state(self, state):
            return [state[k] for k in ('x', 'dx', 'y', 'dy', 'z', 'dz',
                                       'phi', 'dphi', 'theta', 'dtheta', 'psi', 'dpsi')]

        def _get_motors(self, motors):
            return motors
    return HoverRef


def e_series(_Task, Lander, vp):
    rng = np.random.default_rng(777)
    hov = hover_motor(vp)
    MOTORVAL = 1.625e-2  # reference lander.py:21

    HoverRef = make_hover_ref(_Task)

    T = 1100
    ones = np.ones((T, 4))
    cases = {}
    # E01 BASELINE config 1: lander.py constant thrust, exact float64 MOTORVAL
    cases["E01_lander_const_f64"] = run_env(Lander(), MOTORVAL * ones, seed=0)
    # E02 same with float32-representable MOTORVAL (device-comparable)
    cases["E02_lander_const"] = run_env(Lander(), f32r(MOTORVAL) * ones, seed=0)
    # E03 lander.py --random:  MOTORVAL * randn(4)  (negatives clip to 0)
    cases["E03_lander_randn"] = run_env(Lander(), f32r(MOTORVAL * rng.standard_normal((T, 4))), seed=1)
    # E04 uniform [-1,1)^4 actions: tilt limit within a few steps
    for k in range(4):
        cases["E04_lander_uniform_%d" % k] = run_env(
            Lander(), f32r(rng.uniform(-1, 1, (T, 4))), seed=10 + k)
    # E05 roll torque: tilt limit in 3 steps
    cases["E05_lander_roll"] = run_env(Lander(), np.array([0., 1., 1., 0.]) * ones, seed=2)
    # E06 hover thrust with small noise for > 1000 steps: step limit fires
    cases["E06_lander_hover_limit"] = run_env(Lander(), f32r(hov) * ones, seed=3)
    # E07 soft landing from 5 cm: LEVELING -> LANDED -> done + bonus
    cases["E07_lander_soft_landing"] = run_env(Lander(), f32r(1.6e-2) * ones, seed=4, altitude=0.05)
    # E08 out of bounds: sustained lateral tilt then hover
    a = f32r(hov * np.array([0.99, 1.01, 1.01, 0.99])) * ones
    a[40:] = f32r(hov)
    cases["E08_lander_oob"] = run_env(Lander(), a, seed=5)
    # E09 near-hover noisy actions (low-churn benchmark law)
    cases["E09_lander_noisy_hover"] = run_env(
        Lander(), f32r(hov * (1 + 0.01 * rng.standard_normal((T, 4)))), seed=6)
    # E10 float32 action arrays: under NumPy >= 2 promotion the reference then
    #     evaluates the motor model in float32 (documented quirk)
    cases["E10_lander_f32_actions"] = run_env(
        Lander(), f32r(hov * (1 + 0.01 * rng.standard_normal((T, 4)))), seed=7,
        action_dtype=np.float32)
    # Hover3D
    cases["E20_hover_const"] = run_env(HoverRef(), f32r(hov) * ones, seed=8)
    cases["E21_hover_uniform"] = run_env(HoverRef(), f32r(rng.uniform(-1, 1, (T, 4))), seed=9)
    cases["E22_hover_soft_landing"] = run_env(HoverRef(), f32r(1.6e-2) * ones, seed=11, altitude=0.05)
    cases["E23_hover_noisy"] = run_env(
        HoverRef(), f32r(hov * (1 + 0.01 * rng.standard_normal((T, 4)))), seed=12)
    return cases


def r_series(_Task, Lander, vp):
    """Resets to a pose: _Task._reset(pose=(x, y, altitude, phi_deg, theta_deg), perturb=...)
    (task.py:145-188; lander.py:85 is upstream's caller), then ordinary steps."""
    rng = np.random.default_rng(515)
    hov = hover_motor(vp)
    HoverRef = make_hover_ref(_Task)
    T = 400
    ones = np.ones((T, 4))
    noisy = lambda: f32r(hov * (1 + 0.01 * rng.standard_normal((T, 4))))
    cases = {}
    cases["R01_pose_offset_tilted"] = run_env(Lander(), noisy(), seed=80, pose=(3.0, -2.5, 8.0, 10.0, -5.0))
    cases["R02_pose_no_perturb"] = run_env(Lander(), f32r(hov) * ones, seed=81, pose=(-4.0, 1.5, 6.0, -12.0, 20.0),
                                           perturb=False)
    cases["R03_pose_on_ground"] = run_env(Lander(), f32r(1.8e-2) * ones, seed=82, pose=(1.0, 1.0, 0.0, 0.0, 0.0))
    cases["R04_pose_near_limit"] = run_env(Lander(), noisy(), seed=83, pose=(9.5, -9.0, 12.0, 40.0, 0.0), perturb=False)
    cases["R05_pose_past_tilt_limit"] = run_env(Lander(), noisy(), seed=84, pose=(0.0, 0.0, 5.0, 50.0, 0.0))
    cases["R06_hover_pose"] = run_env(HoverRef(), noisy(), seed=85, pose=(2.0, 2.0, 4.0, 5.0, 5.0))
    cases["R07_pose_low_descent"] = run_env(Lander(), f32r(1.6e-2) * ones, seed=86, pose=(0.5, -0.5, 0.25, 0.0, 0.0),
                                            perturb=False)
    return cases


def v_series(_Task, Lander, vp):
    """1D / 2D task variants on the LIVE _Task / Lander: the two hooks of the retired variant
    classes -- _get_motors fan-out and _get_state sub-selection
    (attic/gym_copter/envs/lander1d.py:43-48, lander2d.py:43-50, hover1d.py:44-50,
    hover2d.py:44-50; those files themselves import names that no longer exist) -- plugged into
    the reference's current step()/reset()/reward code."""
    rng = np.random.default_rng(4242)
    hov = hover_motor(vp)
    KEYS = {"1d": ('z', 'dz'), "2d": ('y', 'dy', 'z', 'dz', 'phi', 'dphi')}
    FAN = {"1d": lambda m: [m[0], m[0], m[0], m[0]], "2d": lambda m: [m[0], m[1], m[1], m[0]]}

    def variant(kind, dim):
        keys, fan, nact = KEYS[dim], FAN[dim], {"1d": 1, "2d": 2}[dim]

        class V(Lander):
            TASK_NAME = kind + dim

            def __init__(self):
                _Task.__init__(self, len(keys), nact)
                self.viewer = None

            def _get_state(self, state):
                return [state[k] for k in keys]

            def _get_motors(self, motors):
                return fan(motors)

            if kind == "hover":
                def _get_reward(self, status, state, d, x, y):     # attic hover.py:18-21
                    return 1
        return V()

    T = 1100
    cases = {}
    for dim, nact in (("1d", 1), ("2d", 2)):
        ones = np.ones((T, nact))
        for kind in ("lander", "hover"):
            tag = "V_%s%s" % (kind, dim)
            cases[tag + "_const"] = run_env(variant(kind, dim), f32r(1.625e-2) * ones, seed=30)
            cases[tag + "_uniform"] = run_env(variant(kind, dim), f32r(rng.uniform(-1, 1, (T, nact))), seed=31)
            cases[tag + "_noisy_hover"] = run_env(
                variant(kind, dim), f32r(hov * (1 + 0.01 * rng.standard_normal((T, nact)))), seed=32)
            cases[tag + "_soft_landing"] = run_env(variant(kind, dim), f32r(1.6e-2) * ones, seed=33,
                                                   altitude=0.05)
        # 2D only: differential thrust rolls the copter to the tilt limit / out of bounds
        if dim == "2d":
            a = f32r(hov * np.array([0.99, 1.01])) * ones
            a[40:] = f32r(hov)
            cases["V_lander2d_oob"] = run_env(variant("lander", dim), a, seed=34)
            cases["V_hover2d_roll"] = run_env(variant("hover", dim), np.array([0., 1.]) * ones, seed=35)
    return cases


def w_series(Dynamics, Lander, vp):
    """Other vehicles / worlds on the LIVE code: the reference's Lander flown with a different
    `vehicle_params` dict (the module global that task.py:161 hands to Dynamics) and a different
    gravity constant (Dynamics.G, dynamics/__init__.py:76).  Parameter sets: a heavier airframe,
    a light fast one, and Mars gravity with the weak rotors of attic/mars/dynamics/ingenuity.py
    (G = 3.721).  All values are float32-representable."""
    task_mod = importlib.import_module("gym_copter.envs.task")
    rng = np.random.default_rng(990)
    worlds = {
        "heavy": (dict(B=6.5e-3, D=2.5e-6, M=2.5, L=0.45, Ix=3.0, Iy=2.5, Iz=4.5, Jr=38e-4, maxrpm=12000), 9.80665),
        "light": (dict(B=3.0e-3, D=1.0e-6, M=0.75, L=0.25, Ix=1.0, Iy=1.25, Iz=2.0, Jr=38e-4, maxrpm=18000), 9.80665),
        "mars": (dict(B=2.0e-3, D=2.0e-6, M=1.8, L=0.6, Ix=2.5, Iy=2.0, Iz=3.5, Jr=38e-4, maxrpm=15000), 3.721),
    }
    T = 1100
    cases = {}
    saved_vp, saved_g = task_mod.vehicle_params, Dynamics.G
    try:
        for name, (v, g) in worlds.items():
            v = {k: float(np.float32(val)) for k, val in v.items()}
            g = float(np.float32(g))
            task_mod.vehicle_params = v
            Dynamics.G = g
            hov = hover_motor(v, g)
            extra = dict(vehicle=np.array([v[k] for k in ("B", "D", "M", "L", "Ix", "Iy", "Iz", "Jr", "maxrpm")] + [g]))
            laws = {
                "const": f32r(0.98 * hov) * np.ones((T, 4)),
                "noisy_hover": f32r(hov * (1 + 0.01 * rng.standard_normal((T, 4)))),
                "uniform": f32r(rng.uniform(-1, 1, (T, 4))),
                "yaw_roll": f32r(hov * np.array([1.02, 1.0, 0.99, 0.99])) * np.ones((T, 4)),
            }
            for j, (law, acts) in enumerate(laws.items()):
                c = run_env(Lander(), acts, seed=60 + j)
                c.update(extra)
                cases["W_%s_%s" % (name, law)] = c
            c = run_env(Lander(), f32r(0.97 * hov) * np.ones((T, 4)), seed=70, altitude=0.05)
            c.update(extra)
            cases["W_%s_soft_landing" % name] = c
    finally:
        task_mod.vehicle_params, Dynamics.G = saved_vp, saved_g
    return cases


def save(name, cases):
    flat = {}
    for cname, c in cases.items():
        for k, v in c.items():
            flat["%s/%s" % (cname, k)] = v
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **flat)
    print("wrote %s  (%d cases, %.1f kB)" % (path, len(cases), os.path.getsize(path) / 1e3))


def main():
    Dynamics, vp, _Task, Lander = load_reference()
    print("numpy", np.__version__, "reference", REF)
    save("dynamics_traces.npz", d_series(Dynamics, vp))
    save("env_traces.npz", e_series(_Task, Lander, vp))
    save("pid_traces.npz", p_series(Lander, load_mars_pid(), make_hover_ref(_Task)))
    save("variant_traces.npz", v_series(_Task, Lander, vp))
    save("vehicle_traces.npz", w_series(Dynamics, Lander, vp))
    save("pose_traces.npz", r_series(_Task, Lander, vp))
    save("mars_traces.npz", m_series(*load_mars_dynamics()))
    # known-answer constants observed from the reference (used as spot checks)
    meta = dict(numpy_version=np.array(np.__version__), hover_motor=np.float64(hover_motor(vp)))
    np.savez(os.path.join(OUT, "meta.npz"), **meta)




# --------------------------------------------------------------------------
# P-series: closed loop with the reference's PID heuristic (attic/mars) on the LIVE Lander
# --------------------------------------------------------------------------
def load_mars_pid():
    """attic/mars/pidcontrollers/__init__.py depends on NumPy only."""
    spec = importlib.util.spec_from_file_location(
        "mars_pidcontrollers", os.path.join(REF, "attic", "mars", "pidcontrollers", "__init__.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def run_pid_episode(Lander, pid, seed, steps, altitude=None, gains=None):
    """The heuristic loop of attic/mars/task.py:134-179 with the controller wiring of
    attic/mars/lander3d.py:32-36 + heuristic() :64-87 (restated here because that class needs
    modules that no longer import), driving the live reference Lander.  The observation is
    handed to the controllers as Python floats and the action is rounded to float32 before
    env.step (the action space is float32), so a device path can be fed identical numbers."""
    gains = gains or {}
    rate = dict(Kp=1.0, Ki=0, Kd=1)
    rate.update(gains.get("rate", {}))
    pos = dict(Kp=0.00001, Ki=0.1, Kd=4, target=0)
    pos.update(gains.get("pos", {}))
    des = dict(Kp=1.15, Kd=1.33)
    des.update(gains.get("descent", {}))
    phi_rate_pid = pid.AngularVelocityPidController(**rate)
    theta_rate_pid = pid.AngularVelocityPidController(**rate)
    x_poshold_pid = pid.PositionHoldPidController(**pos)
    y_poshold_pid = pid.PositionHoldPidController(**pos)
    descent_pid = pid.DescentPidController(**des)

    env = Lander()
    if altitude is not None:
        env.set_altitude(altitude)
    np.random.seed(seed)
    obs, _ = env.reset()
    d = env.dynamics
    force = f32r(d._perturb * d.M)
    d.perturb(force.copy())
    rec = dict(obs=[], reward=[], done=[], action=[], x=[])
    for t in range(steps):
        x, dx, y, dy, z, dz, phi, dphi, theta, dtheta = [float(v) for v in obs]
        phi_rate_todo = phi_rate_pid.getDemand(dphi)
        y_pos_todo = x_poshold_pid.getDemand(y, dy)
        phi_todo = phi_rate_todo + y_pos_todo
        theta_rate_todo = theta_rate_pid.getDemand(-dtheta)
        x_pos_todo = y_poshold_pid.getDemand(x, dx)
        theta_todo = theta_rate_todo + x_pos_todo
        descent_todo = descent_pid.getDemand(z, dz)
        tt, r, p = (descent_todo + 1) / 2, phi_todo, theta_todo
        action = f32r([tt - r - p, tt + r + p, tt + r - p, tt - r + p])
        obs, reward, done, _, _ = env.step(action)
        rec["obs"].append(obs)
        rec["reward"].append(float(reward))
        rec["done"].append(bool(done))
        rec["action"].append(action)
        rec["x"].append(d._x.copy())
        if done:
            break
    return dict(seed=np.int64(seed), altitude=np.float64(env.initial_altitude), force=force[:3],
                obs=np.asarray(rec["obs"], dtype=np.float32), reward=np.asarray(rec["reward"]),
                done=np.asarray(rec["done"]), action=np.asarray(rec["action"]), x=np.asarray(rec["x"]),
                rate_gains=np.array([rate["Kp"], rate["Ki"], rate["Kd"]], dtype=np.float64),
                pos_gains=np.array([pos["Kp"], pos["Ki"], pos["Kd"], pos["target"]], dtype=np.float64),
                descent_gains=np.array([des["Kp"], des["Kd"]], dtype=np.float64))


def run_pid_hover_episode(HoverRef, pid, seed, steps, altitude=None, gains=None):
    """The hover heuristic of attic/mars/hover3d.py:65-92 (roll / pitch / yaw rate controllers, two
    position-hold controllers, the altitude-hold controller of attic/mars/hover.py:23) with the
    reference's controller classes, driving the Hover3D restatement on the live _Task."""
    gains = gains or {}
    rate = dict(Kp=1.0, Ki=0, Kd=1)
    rate.update(gains.get("rate", {}))
    pos = dict(Kp=0.00001, Ki=0.1, Kd=4, target=0)
    pos.update(gains.get("pos", {}))
    alt = dict(Kp=0.2, Ki=3, Kd=0, target=5)
    alt.update(gains.get("alt", {}))
    roll_rate_pid = pid.AngularVelocityPidController(**rate)
    pitch_rate_pid = pid.AngularVelocityPidController(**rate)
    yaw_rate_pid = pid.AngularVelocityPidController(**rate)
    x_poshold_pid = pid.PositionHoldPidController(**pos)
    y_poshold_pid = pid.PositionHoldPidController(**pos)
    altpid = pid.AltitudeHoldPidController(**alt)

    env = HoverRef()
    if altitude is not None:
        env.set_altitude(altitude)
    np.random.seed(seed)
    obs, _ = env.reset()
    d = env.dynamics
    force = f32r(d._perturb * d.M)
    d.perturb(force.copy())
    rec = dict(obs=[], reward=[], done=[], action=[], x=[])
    for t in range(steps):
        x, dx, y, dy, z, dz, phi, dphi, theta, dtheta, _, dpsi = [float(v) for v in obs]
        roll_rate_todo = roll_rate_pid.getDemand(dphi)
        y_pos_todo = x_poshold_pid.getDemand(y, dy)
        pitch_rate_todo = pitch_rate_pid.getDemand(-dtheta)
        x_pos_todo = y_poshold_pid.getDemand(x, dx)
        roll_todo = roll_rate_todo + y_pos_todo
        pitch_todo = pitch_rate_todo + x_pos_todo
        yaw_todo = yaw_rate_pid.getDemand(-dpsi)
        hover_todo = altpid.getDemand(z, dz)
        tt, r, p, yw = (hover_todo + 1) / 2, roll_todo, pitch_todo, yaw_todo
        action = f32r([tt - r - p - yw, tt + r + p - yw, tt + r - p + yw, tt - r + p + yw])
        obs, reward, done, _, _ = env.step(action)
        rec["obs"].append(obs)
        rec["reward"].append(float(reward))
        rec["done"].append(bool(done))
        rec["action"].append(action)
        rec["x"].append(d._x.copy())
        if done:
            break
    return dict(seed=np.int64(seed), altitude=np.float64(env.initial_altitude), force=force[:3],
                heuristic=np.array("hover"),
                obs=np.asarray(rec["obs"], dtype=np.float32), reward=np.asarray(rec["reward"]),
                done=np.asarray(rec["done"]), action=np.asarray(rec["action"]), x=np.asarray(rec["x"]),
                rate_gains=np.array([rate["Kp"], rate["Ki"], rate["Kd"]], dtype=np.float64),
                pos_gains=np.array([pos["Kp"], pos["Ki"], pos["Kd"], pos["target"]], dtype=np.float64),
                alt_gains=np.array([alt["Kp"], alt["Ki"], alt["Kd"], alt["target"]], dtype=np.float64))


def p_series(Lander, pid, HoverRef=None):
    cases = {}
    # reference gains (tuned upstream for the retired mars dynamics: bang-bang on the live model)
    cases["P01_pid_default"] = run_pid_episode(Lander, pid, seed=20, steps=400)
    cases["P02_pid_default_low"] = run_pid_episode(Lander, pid, seed=21, steps=400, altitude=2.0)
    # gains scaled to the live vehicle (hover motor value 0.01656): a descent that lands
    soft = dict(descent=dict(Kp=0.004, Kd=0.012), rate=dict(Kp=0.002, Kd=0.002),
                pos=dict(Kp=0.0002, Ki=0.0, Kd=0.0))
    cases["P03_pid_soft"] = run_pid_episode(Lander, pid, seed=22, steps=1100, gains=soft)
    cases["P04_pid_soft"] = run_pid_episode(Lander, pid, seed=23, steps=1100, gains=soft)
    if HoverRef is not None:
        # the hover heuristic (attic/mars/hover3d.py:65-92): upstream's gains, then gains that do hover
        cases["H01_hover_default"] = run_pid_hover_episode(HoverRef, pid, seed=40, steps=400)
        cases["H02_hover_default_low"] = run_pid_hover_episode(HoverRef, pid, seed=41, steps=400, altitude=3.0)
        tuned = dict(alt=dict(Kp=0.02, Ki=5.0, Kd=0, target=5), rate=dict(Kp=0.002, Kd=0.002),
                     pos=dict(Kp=0.0002, Ki=0.0, Kd=0.0))
        cases["H03_hover_tuned"] = run_pid_hover_episode(HoverRef, pid, seed=42, steps=1100, gains=tuned)
        cases["H04_hover_tuned"] = run_pid_hover_episode(HoverRef, pid, seed=43, steps=1100, gains=tuned, altitude=5.0)
    return cases


# --------------------------------------------------------------------------
# M-series: the retired Mars dynamics (lift-coefficient thrust law, air density, rotor-inertia term)
# --------------------------------------------------------------------------
def load_mars_dynamics():
    """attic/mars/dynamics/__init__.py and ingenuity.py depend on NumPy only; ingenuity.py imports its
    base class as `from dynamics import MultirotorDynamics`, so the package is registered under that name."""
    d = os.path.join(REF, "attic", "mars", "dynamics")
    spec = importlib.util.spec_from_file_location("dynamics", os.path.join(d, "__init__.py"),
                                                  submodule_search_locations=[d])
    pkg = importlib.util.module_from_spec(spec)
    saved = sys.modules.get("dynamics")
    sys.modules["dynamics"] = pkg
    try:
        spec.loader.exec_module(pkg)
        ing = importlib.import_module("dynamics.ingenuity")
    finally:
        if saved is None:
            sys.modules.pop("dynamics", None)
        else:
            sys.modules["dynamics"] = saved
    return pkg.MultirotorDynamics, ing.CoaxialDynamics, ing.IngenuityDynamics


def run_mars(dyn, x0, force, motors):
    """motors [T,4]: setMotors(m); update() per tick (attic/mars/task.py's loop)."""
    dyn.setState(np.array(x0, dtype=np.float64))
    status0 = dyn.getStatus()
    if force is not None:
        dyn.perturb(np.array(force, dtype=np.float64))
    T = motors.shape[0]
    xs = np.zeros((T, 12))
    st = np.zeros(T, dtype=np.int8)
    for t in range(T):
        dyn.setMotors(motors[t])
        dyn.update()
        xs[t] = dyn._x
        st[t] = dyn.getStatus()
    vehicle = np.array([dyn.B, dyn.D, dyn.M, dyn.L, dyn.Ix, dyn.Iy, dyn.Iz, dyn.Jr, dyn.maxrpm,
                        dyn.G, dyn.rho, dyn.C_L], dtype=np.float64)
    return dict(fps=np.int64(round(1.0 / dyn._dt)), x0=np.array(x0, dtype=np.float64), status0=np.int8(status0),
                force=np.zeros(6) if force is None else np.array(force, dtype=np.float64),
                motors=motors, x=xs, status=st, vehicle=vehicle)


def m_series(MultirotorDynamics, CoaxialDynamics, IngenuityDynamics):
    rng = np.random.default_rng(4242)
    cases = {}

    def lift_hover(d):      # 4 * 0.5*rho*S*C_L*(w*L/2)^2 = M*G
        kl = 0.5 * d.rho * d.S * d.C_L * (d.L / 2) ** 2
        return float(np.sqrt(d.M * d.G / (4 * kl)) / (d.maxrpm * np.pi / 30))

    def airborne(alt=10.0, **kw):
        x = np.zeros(12)
        x[4] = -alt
        for k, v in kw.items():
            x[int(k[1:])] = v
        return x

    T = 1000
    ones = np.ones((T, 4))
    # Ingenuity on Mars (ingenuity.py:46-75): thin air, the hover motor value is ~0.73
    d = IngenuityDynamics(100)
    hov = lift_hover(d)
    cases["M01_ingenuity_hover"] = run_mars(d, airborne(), f32r([4.5, -2.25, 1.0, 0, 0, 0]), f32r(hov) * ones)
    d = IngenuityDynamics(100)
    m = f32r(hov * np.array([1.0, 1.004, 0.998, 1.002]))          # all three torques and Omega != 0
    cases["M02_ingenuity_asymmetric"] = run_mars(d, airborne(50.0, i7=0.02, i9=-0.01, i11=0.05), f32r([1, 2, 3, 0, 0, 0]), m * ones)
    d = IngenuityDynamics(100)
    mseq = f32r(np.clip(hov * (1.0 + 0.03 * rng.standard_normal((T, 4))), 0, 1))
    cases["M03_ingenuity_random"] = run_mars(
        d, airborne(200.0, i1=0.5, i3=-0.25, i6=0.05, i7=0.3, i8=-0.03, i9=-0.2, i10=0.3, i11=-0.4),
        f32r([9.5, -9.5, 3.0, 0, 0, 0]), mseq)
    # the same airframe in Earth's air and gravity (the module's default world parameters)
    vp = dict(B=5.e-6, D=2.e-6, M=1.380, L=0.350, C_L=0.4, Ix=2, Iy=2, Iz=3, Jr=38e-4, maxrpm=15000)
    d = CoaxialDynamics(vp, 100, {'G': 9.80655, 'rho': 1.225})
    hov_e = lift_hover(d)
    cases["M04_earth_hover"] = run_mars(d, airborne(), f32r([-3.5, 21.0, -9.75, 0, 0, 0]), f32r(hov_e) * ones)
    # a heavier rotor (Jr x 5) and asymmetric inertia: the rotor-inertia term carries weight
    vp2 = dict(vp, Jr=0.019, Ix=1.5, Iy=2.5, C_L=0.55, L=0.3)
    d = CoaxialDynamics(vp2, 100, {'G': 3.721, 'rho': 0.25})
    hov2 = lift_hover(d)
    mseq = f32r(np.clip(hov2 * (1.0 + 0.02 * rng.standard_normal((500, 4))), 0, 1))
    cases["M05_heavy_rotor"] = run_mars(d, airborne(300.0, i7=0.1, i9=0.08, i11=-0.1), None, mseq)
    # take-off from LANDED, free-fall crash, soft landing: the flight-status machine of update()
    d = IngenuityDynamics(100)
    ramp = f32r(np.linspace(0.0, min(1.0, 1.3 * hov), 300))[:, None] * np.ones((300, 4))
    cases["M06_takeoff"] = run_mars(d, np.zeros(12), None, ramp)
    d = IngenuityDynamics(100)
    cases["M07_crash"] = run_mars(d, airborne(3.0), None, np.zeros((300, 4)))
    d = IngenuityDynamics(100)
    cases["M08_soft_landing"] = run_mars(d, airborne(0.05, i6=0.01), None, f32r(0.97 * hov) * np.ones((400, 4)))
    return cases


if __name__ == "__main__":
    main()
