"""
oracle/refcpu.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Scalar float64 CPU restatement of the gym-copter hot path, one environment per
object, written so that every float64 operation happens in the same order as in
the reference (so results are bit-identical to it, see tests/test_oracle_golden.py,
which pins this file against golden traces captured from the real reference by
tests/golden/generate_golden.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product package (gym_copter_amd) never does.

Reference lines restated here (paths relative to the upstream checkout):
  gym_copter/dynamics/__init__.py:114-197   Dynamics.setMotors      -> RigidBody.set_motors
  gym_copter/dynamics/__init__.py:249-290   _computeStateDerivative -> RigidBody._derivative
  gym_copter/dynamics/__init__.py:292-302   _bodyZToInertial        -> RigidBody._thrust_ned
  gym_copter/dynamics/__init__.py:210-217   setState                -> RigidBody.set_state
  gym_copter/dynamics/__init__.py:227-229   perturb                 -> RigidBody.perturb
  gym_copter/dynamics/vehicles/dji_phantom.py:9-26                  -> DJI_PHANTOM
  gym_copter/envs/task.py:77-137            _Task.step              -> TaskOracle.step
  gym_copter/envs/task.py:145-197           _Task._reset            -> TaskOracle.reset
  gym_copter/envs/lander.py:46-74           Lander._get_reward      -> TaskOracle._lander_reward
  attic/gym_copter/envs/hover.py:18-21      _Hover._get_reward      -> task == "hover3d"
  attic/gym_copter/envs/hover3d.py:32-37    Hover3D._get_state      -> 12-component observation
"""

from dataclasses import dataclass

import numpy as np

# flight status codes (dynamics/__init__.py:65-68)
CRASHED, LANDED, LEVELING, AIRBORNE = 0, 1, 2, 3

# state slots (dynamics/__init__.py:48-59)
X, DX, Y, DY, Z, DZ, PHI, DPHI, THETA, DTHETA, PSI, DPSI = range(12)

G = 9.80665                      # dynamics/__init__.py:76
LANDING_VEL_X = 2.0              # :71
LANDING_VEL_Y = 1.0              # :72
LANDING_ANGLE = np.pi / 4        # :73


@dataclass(frozen=True)
class VehicleParams:
    B: float
    D: float
    M: float
    L: float
    Ix: float
    Iy: float
    Iz: float
    Jr: float
    maxrpm: float


# dji_phantom.py:9-26.  Ix/Iy/Iz/maxrpm are Python ints upstream; the values
# (and every product/quotient formed from them) are identical as floats.
DJI_PHANTOM = VehicleParams(B=5.e-3, D=2.e-6, M=1.380, L=0.350,
                            Ix=2, Iy=2, Iz=3, Jr=38e-4, maxrpm=15000)


@dataclass(frozen=True)
class TaskParams:
    """_Task constructor keywords (task.py:32-38) + Lander constants (lander.py:17-23)."""
    initial_random_force: float = 30
    out_of_bounds_penalty: float = 100
    max_steps: int = 1000
    max_angle: float = 45          # degrees
    bounds: float = 10
    initial_altitude: float = 10
    frames_per_second: int = 100   # task.py:25
    target_radius: float = 2
    yaw_penalty_factor: float = 50
    xyz_penalty_factor: float = 25
    dz_max: float = 10
    dz_penalty: float = 100
    inside_radius_bonus: float = 100


class RigidBody:
    """One quad-X rigid body, forward-Euler integrated in Euler angles."""

    def __init__(self, vp=DJI_PHANTOM, frames_per_second=100, g=G, mars=None):
        """mars = (rho, C_L): the retired Mars model instead of the live one -- lift-coefficient
        thrust law and a live rotor-inertia term (attic/mars/dynamics/__init__.py:78-164)."""
        self.vp = vp
        self.mars = mars
        self.g = g                       # Dynamics.G (a class constant upstream, :76)
        self.dt = 1. / frames_per_second
        self.ticks = 0
        self.x = np.zeros(12)
        self.dxdt = np.zeros(12)
        self.status = LANDED
        self.pending = np.zeros(6)       # stored as force / M

    def set_state(self, x):
        self.x = np.array(x)
        self.status = AIRBORNE if self.x[Z] < 0 else LANDED

    def perturb(self, force):
        self.pending = force / self.vp.M

    @staticmethod
    def _thrust_ned(body_z, phi, theta, psi):
        cph, cth, cps = np.cos(phi), np.cos(theta), np.cos(psi)
        sph, sth, sps = np.sin(phi), np.sin(theta), np.sin(psi)
        col = np.array([sph * sps + cph * cps * sth,
                        cph * sps * sth - cps * sph,
                        cph * cth])
        return body_z * col

    def _derivative(self, acc, netz, U2, U3, U4, Omega):
        p = self.vp
        x, k, out = self.x, self.pending, self.dxdt
        dphi, dthe, dpsi = x[DPHI], x[DTHETA], x[DPSI]
        out[X] = x[DX]
        out[DX] = acc[0] + k[0]
        out[Y] = x[DY]
        out[DY] = acc[1] + k[1]
        out[Z] = x[DZ]
        out[DZ] = netz + k[2]
        out[PHI] = dphi
        out[DPHI] = (dpsi * dthe * (p.Iy - p.Iz) / p.Ix - p.Jr / p.Ix * dthe * Omega
                     + U2 / p.Ix + k[3])
        out[THETA] = dthe
        out[DTHETA] = (-(dpsi * dphi * (p.Iz - p.Ix) / p.Iy + p.Jr / p.Iy * dphi * Omega
                         + U3 / p.Iy) + k[4])
        out[PSI] = dpsi
        out[DPSI] = dthe * dphi * (p.Ix - p.Iy) / p.Iz + U4 / p.Iz + k[5]

    def set_motors(self, motorvals):
        p = self.vp
        w = np.array(motorvals) * p.maxrpm * np.pi / 30       # rad/s
        w2 = w ** 2
        if self.mars is None:
            U1 = p.B * np.sum(w2)
            U2 = p.L * p.B * ((w2[1] + w2[2]) - (w2[0] + w2[3]))  # roll right
            U3 = p.L * p.B * ((w2[1] + w2[3]) - (w2[0] + w2[2]))  # pitch forward
            U4 = p.D * ((w2[0] + w2[1]) - (w2[2] + w2[3]))        # yaw cw
            Omega = 0                                             # rotor-inertia term disabled upstream
        else:
            # attic/mars/dynamics/__init__.py:135-164 (setMotors) with S from :88
            rho, C_L = self.mars
            S = .05 * p.L * 4
            Omega = (w[0] + w[1]) - (w[2] + w[3])                 # u4(omegas), before squaring
            velocity = w * p.L / 2
            lift = 0.5 * rho * S * C_L * (velocity ** 2)
            U1 = np.sum(lift)
            U2 = (lift[1] + lift[2]) - (lift[0] + lift[3])
            U3 = (lift[1] + lift[3]) - (lift[0] + lift[2])
            U4 = p.D * ((w2[0] + w2[1]) - (w2[2] + w2[3]))

        acc = self._thrust_ned(-U1 / p.M, self.x[PHI], self.x[THETA], self.x[PSI])
        netz = acc[2] + self.g

        if self.status == LANDED and netz < 0:
            self.status = AIRBORNE

        if self.status == LEVELING:
            self.x[PHI] = 0
            self.x[THETA] = 0
            self.status = LANDED
        elif self.status == AIRBORNE:
            if self.x[Z] > 0 and self.x[DZ] > 0:
                # ground contact: freeze (no integrate, no tick, perturbation kept)
                hard = (self.x[DZ] > LANDING_VEL_Y or abs(self.x[DY]) > LANDING_VEL_X
                        or abs(self.x[PHI]) > LANDING_ANGLE)
                self.status = CRASHED if hard else LEVELING
                return
            self._derivative(acc, netz, U2, U3, U4, Omega)
            self.dxdt[1::2] += self.pending          # second application (upstream behaviour)
            self.x += self.dt * self.dxdt

        self.pending = np.zeros(6)
        self.ticks += 1


# task -> (reward kind, first observed state slot, observation size, motor fan-out).
# 3D: lander.py:39-44 / attic hover3d.py:32-37.  1D / 2D: the _get_state sub-selection and
# _get_motors fan-out of attic/gym_copter/envs/lander1d.py:43-48, lander2d.py:43-50,
# hover1d.py:44-50, hover2d.py:44-50 plugged into the live _Task hooks (task.py:94, :133).
TASKS = {
    "lander3d": ("lander", 0, 10, (0, 1, 2, 3)),
    "hover3d": ("hover", 0, 12, (0, 1, 2, 3)),
    "lander2d": ("lander", 2, 6, (0, 1, 1, 0)),     # obs (y,dy,z,dz,phi,dphi), motors [m0,m1,m1,m0]
    "lander1d": ("lander", 4, 2, (0, 0, 0, 0)),     # obs (z,dz),               motors [m0,m0,m0,m0]
    "hover2d": ("hover", 2, 6, (0, 1, 1, 0)),
    "hover1d": ("hover", 4, 2, (0, 0, 0, 0)),
}


def task_action_dim(task):
    return max(TASKS[task][3]) + 1


class TaskOracle:
    """Single-environment Lander / Hover task (3D, or a 2D / 1D variant) around one RigidBody."""

    def __init__(self, task="lander3d", tp=TaskParams(), vp=DJI_PHANTOM, substeps=1,
                 action_dtype_passthrough=False, g=G):
        assert task in TASKS
        self.task, self.tp, self.vp, self.g = task, tp, vp, g
        self.kind, self.obs_first, self.obs_dim, self.fan = TASKS[task]
        self.act_dim = task_action_dim(task)
        self.substeps = substeps
        self.passthrough = action_dtype_passthrough
        self.max_angle = np.radians(tp.max_angle)
        self.body = None

    # -- reset (task.py:145-197) ------------------------------------------
    def reset(self, force_xyz=None, rng=None, pose=None, perturb=True):
        """pose = (x, y, altitude, phi_deg, theta_deg) and perturb: _Task._reset's keywords
        (task.py:145, :149-150, :163-170, :176)."""
        tp = self.tp
        self.prev_shaping = None
        self.body = RigidBody(self.vp, tp.frames_per_second * self.substeps, self.g)
        if pose is None:
            pose = (0, 0, tp.initial_altitude, 0, 0)
        x0 = np.zeros(12)
        x0[X] = pose[0]
        x0[Y] = pose[1]
        x0[Z] = -pose[2]
        x0[PHI] = np.radians(pose[3])
        x0[THETA] = np.radians(pose[4])
        self.body.set_state(x0)
        if not perturb:
            force_xyz = [0, 0, 0]
        elif force_xyz is None:
            draw = (rng or np.random).uniform
            force_xyz = [draw(-tp.initial_random_force, +tp.initial_random_force) for _ in range(3)]
        self.force = np.array([force_xyz[0], force_xyz[1], force_xyz[2], 0, 0, 0], dtype=np.float64)
        if perturb:
            self.body.perturb(self.force)
        self.steps = 0
        return self.step(np.zeros(4), initializing=True)[0]

    # -- rewards ----------------------------------------------------------
    def _lander_reward(self, status0, x):
        tp = self.tp
        pos = np.array([x[X], x[DX], x[Y], x[DY], x[Z], x[DZ]])
        yaw = np.array([x[PSI], x[DPSI]])
        shaping = -(tp.xyz_penalty_factor * np.sqrt(np.sum(pos ** 2)) +
                    tp.yaw_penalty_factor * np.sqrt(np.sum(yaw ** 2)))
        if abs(x[DZ]) > tp.dz_max:
            shaping -= tp.dz_penalty
        reward = (shaping - self.prev_shaping) if self.prev_shaping is not None else 0
        self.prev_shaping = shaping
        if status0 == LANDED:
            self.done = True
            if np.sqrt(x[X] ** 2 + x[Y] ** 2) < tp.target_radius:
                reward += tp.inside_radius_bonus
        return reward

    # -- step (task.py:77-137) --------------------------------------------
    def step(self, action, initializing=False):
        tp, b = self.tp, self.body
        status0 = b.status
        if status0 != LANDED:
            if not self.passthrough:
                action = np.asarray(action, dtype=np.float64)
            motors = np.clip(action, 0, 1)
            if not initializing:
                if self.act_dim != 4:
                    motors = [motors[j] for j in self.fan]        # _get_motors
                for _ in range(self.substeps):
                    b.set_motors(motors)
        x = b.x
        self.done = False
        if self.kind == "lander":
            reward = self._lander_reward(status0, x)
        else:
            reward = 1
        if abs(x[X]) >= tp.bounds or abs(x[Y]) >= tp.bounds:
            self.done = True
            reward -= tp.out_of_bounds_penalty
        elif abs(x[PHI]) >= self.max_angle or abs(x[THETA]) >= self.max_angle:
            self.done = True
            reward = -tp.out_of_bounds_penalty
        elif status0 == CRASHED:
            self.done = True
        if self.steps == tp.max_steps:
            self.done = True
        self.steps += 1
        obs = np.array(x[self.obs_first:self.obs_first + self.obs_dim], dtype=np.float32)
        return obs, reward, self.done, False, {}


# ---------------------------------------------------------------------------------------
# PID heuristic policy ("next" row N1).  Restates the retired upstream controllers
#   attic/mars/pidcontrollers/__init__.py:12-146  (_PidController, _SetPointPidController,
#                                                  PositionHold / Descent / AngularVelocity)
#   attic/mars/lander3d.py:32-36 (wiring), :64-87 (heuristic + mixer)
# in float64, one env per object.  Pinned bit-for-bit (actions and trajectories) against
# tests/golden/pid_traces.npz, which was produced by the real controller classes driving the
# real live Lander (tests/golden/generate_golden.py: run_pid_episode).
# ---------------------------------------------------------------------------------------
@dataclass(frozen=True)
class PidGains:
    rate_kp: float = 1.0          # AngularVelocityPidController(Kp=1.0, Ki=0, Kd=1)
    rate_ki: float = 0.0
    rate_kd: float = 1.0
    rate_windup: float = 6.0      # WINDUP_MAX
    rate_big: float = 40.0        # BIG_DEGREES_PER_SECOND
    pos_kp: float = 0.00001       # PositionHoldPidController(Kp=0.00001, Ki=0.1, Kd=4, target=0)
    pos_ki: float = 0.1
    pos_kd: float = 4.0
    pos_target: float = 0.0
    pos_windup: float = 0.2       # _PidController default windup_max
    descent_kp: float = 1.15      # DescentPidController(Kp=1.15, Kd=1.33)
    descent_kd: float = 1.33
    # "hover" heuristic (attic/mars/hover3d.py:65-92): a yaw-rate controller with the rate gains and
    # the altitude-hold controller of attic/mars/hover.py:23 replace the descent law
    heuristic: str = "lander"
    alt_kp: float = 0.2           # AltitudeHoldPidController(Kp=0.2, Ki=3, Kd=0, target=5)
    alt_ki: float = 3.0
    alt_kd: float = 0.0
    alt_target: float = 5.0
    alt_windup: float = 0.2


class _Pid:
    def __init__(self, kp, ki, kd, windup):
        self.kp, self.ki, self.kd, self.windup = kp, ki, kd, windup
        self.last = 0
        self.err_i = 0
        self.d1 = 0
        self.d2 = 0

    def compute(self, target, actual):
        error = target - actual
        pterm = error * self.kp
        iterm = 0
        if self.ki > 0:
            v = self.err_i + error
            self.err_i = -self.windup if v < -self.windup else (self.windup if v > self.windup else v)
            iterm = self.err_i * self.ki
        dterm = 0
        if self.kd > 0:
            de = error - self.last
            dterm = (self.d1 + self.d2 + de) * self.kd
            self.d2 = self.d1
            self.d1 = de
            self.last = error
        return pterm + iterm + dterm


class PidHeuristic:
    """obs[10] (as Python floats) -> 4 motor demands, with the controllers' internal state."""

    def __init__(self, g=PidGains()):
        self.g = g
        self.rate_phi = _Pid(g.rate_kp, g.rate_ki, g.rate_kd, g.rate_windup)
        self.rate_theta = _Pid(g.rate_kp, g.rate_ki, g.rate_kd, g.rate_windup)
        self.pos_for_roll = _Pid(g.pos_kp, g.pos_ki, g.pos_kd, g.pos_windup)    # upstream's x_poshold_pid, fed y
        self.pos_for_pitch = _Pid(g.pos_kp, g.pos_ki, g.pos_kd, g.pos_windup)   # upstream's y_poshold_pid, fed x
        self.rate_psi = _Pid(g.rate_kp, g.rate_ki, g.rate_kd, g.rate_windup)    # hover heuristic only
        self.alt = _Pid(g.alt_kp, g.alt_ki, g.alt_kd, g.alt_windup)             # hover heuristic only
        self.big = np.radians(g.rate_big)

    def _rate(self, pid, w):
        if abs(w) > self.big:          # AngularVelocityPidController.getDemand: reset()
            pid.err_i = 0
            pid.last = 0
        return pid.compute(0, w)

    def _pos(self, pid, x, dx):
        target_velocity = (self.g.pos_target - x) * 1      # posPid = _PidController(1, 0, 0)
        return pid.compute(target_velocity, dx)

    def action(self, obs):
        x, dx, y, dy, z, dz, phi, dphi, theta, dtheta = [float(v) for v in obs[:10]]
        phi_todo = self._rate(self.rate_phi, dphi) + self._pos(self.pos_for_roll, y, dy)
        theta_todo = self._rate(self.rate_theta, -dtheta) + self._pos(self.pos_for_pitch, x, dx)
        if self.g.heuristic == "hover":            # attic/mars/hover3d.py:65-92
            dpsi = float(obs[11])
            yaw_todo = self._rate(self.rate_psi, -dpsi)
            target_velocity = (self.g.alt_target - (-z)) * 1          # AltitudeHold: NED negated
            hover_todo = self.alt.compute(target_velocity, -dz)
            t, r, p, yw = (hover_todo + 1) / 2, phi_todo, theta_todo, yaw_todo
            return np.array([t - r - p - yw, t + r + p - yw, t + r - p + yw, t - r + p + yw])
        descent_todo = z * self.g.descent_kp + dz * self.g.descent_kd
        t, r, p = (descent_todo + 1) / 2, phi_todo, theta_todo
        return np.array([t - r - p, t + r + p, t + r - p, t - r + p])
