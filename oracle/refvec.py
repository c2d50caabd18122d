"""
oracle/refvec.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

NumPy-vectorised batch restatement of the gym-copter hot path.  It follows the
same float64 operation order as oracle/refcpu.py (which is pinned bit-for-bit
against golden traces of the real reference) and tests/test_oracle_vec.py
proves the two agree bit-for-bit, so this file inherits that pin.  On top of
the reference semantics it models the *batch* features the device library adds
(none of which exist upstream): struct-of-arrays state, storage dtype
(float32 state words with float64 arithmetic), inner substeps, masked
auto-reset with a counter-based Philox2x32-10 perturbation draw keyed by
(seed, global env id, episode number), and the
time-limit-as-truncation option.  It is the checker the GPU parity tests and
__graft_entry__.smoke() compare the HIP kernels against, and the
"vectorised" row of bench.py's cpu_baseline.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.

Reference lines followed (paths relative to the upstream checkout):
  gym_copter/dynamics/__init__.py:114-197, :249-302   (physics, via refcpu.RigidBody)
  gym_copter/envs/task.py:77-137, :145-202            (step / reset)
  gym_copter/envs/lander.py:46-74                     (Lander reward)
  attic/gym_copter/envs/hover.py:18-21, hover3d.py:32-37
"""

import numpy as np

from .refcpu import (AIRBORNE, CRASHED, DJI_PHANTOM, G, LANDED, LANDING_ANGLE,
                     LANDING_VEL_X, LANDING_VEL_Y, LEVELING, TASKS, TaskParams, task_action_dim)

AUTORESET_DISABLED, AUTORESET_NEXT_STEP, AUTORESET_SAME_STEP = 0, 1, 2

# ---------------------------------------------------------------------------
# Philox2x32-10 (Salmon et al., SC'11): 64-bit counter, 32-bit key.
# ---------------------------------------------------------------------------
_PH_M = np.uint64(0xD256D193)
_PH_W = np.uint32(0x9E3779B9)
_U32 = np.uint64(0xFFFFFFFF)


def philox2x32_10(c0, c1, key):
    """All arguments uint32 arrays (broadcastable).  Returns 2 uint32 arrays."""
    c0 = np.asarray(c0, dtype=np.uint32)
    c1 = np.asarray(c1, dtype=np.uint32)
    key = np.asarray(key, dtype=np.uint32)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p = _PH_M * c0.astype(np.uint64)
            hi, lo = (p >> np.uint64(32)).astype(np.uint32), (p & _U32).astype(np.uint32)
            c0, c1 = hi ^ key ^ c1, lo
            key = (key + _PH_W).astype(np.uint32)
    return c0, c1


def splitmix64(z):
    """The seed mix behind both Philox keys (every bit of the 64-bit seed matters)."""
    m = (1 << 64) - 1
    z = (int(z) + 0x9E3779B97F4A7C15) & m
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & m
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & m
    return z ^ (z >> 31)


def draw_forces(seed, env_ids, episode, magnitude):
    """Perturbation force draw shared (by specification) with the device kernel.

    counter = (global env id, episode number of that env), key = lo32(splitmix64(seed));
    the 64 output bits give three 21-bit uniforms u = bits * 2^-21 in [0,1);
    F = u * (2*magnitude) - magnitude, float64, un-fused multiply then add.
    Returns [3, n] float64.
    """
    env_ids = np.asarray(env_ids, dtype=np.uint64)
    assert np.all(env_ids < (1 << 32))
    episode = np.broadcast_to(np.asarray(episode, dtype=np.uint32), env_ids.shape)
    key = np.uint32(splitmix64(int(seed) & ((1 << 64) - 1)) & 0xFFFFFFFF)
    r0, r1 = philox2x32_10(env_ids.astype(np.uint32), episode, key)
    bits = (r0 >> np.uint32(11), r1 >> np.uint32(11),
            ((r0 & np.uint32(0x7FF)) << np.uint32(10)) | (r1 & np.uint32(0x3FF)))
    out = np.empty((3, env_ids.shape[0]))
    for i in range(3):
        u = bits[i].astype(np.float64) * (2.0 ** -21)
        out[i] = u * (2.0 * float(magnitude)) - float(magnitude)
    return out


def draw_actions(seed, env_ids, episode, steps, act_dim=4):
    """On-device random policy, shared (by specification) with the K-step kernel:
    counter = (global env id, episode number), key = hi32(splitmix64(seed)) +
    step counter of the episode; the 64 output bits give four 16-bit uniforms
    a = bits * 2^-15 - 1 in [-1, 1), exact in float32; the first `act_dim` are the action.
    Returns [n, act_dim] float32."""
    env_ids = np.asarray(env_ids, dtype=np.uint64)
    episode = np.broadcast_to(np.asarray(episode, dtype=np.uint32), env_ids.shape)
    steps = np.broadcast_to(np.asarray(steps).astype(np.uint32), env_ids.shape)
    with np.errstate(over="ignore"):
        key = (np.uint32(splitmix64(int(seed) & ((1 << 64) - 1)) >> 32) + steps).astype(np.uint32)
    r0, r1 = philox2x32_10(env_ids.astype(np.uint32), episode, key)
    bits = np.stack([r0 >> np.uint32(16), r0 & np.uint32(0xFFFF), r1 >> np.uint32(16), r1 & np.uint32(0xFFFF)], axis=1)
    a = bits.astype(np.float32) * np.float32(2.0 ** -15) - np.float32(1.0)
    return np.ascontiguousarray(a[:, :act_dim])


# ---------------------------------------------------------------------------
# Stored-word rounding shared (by specification) with the device kernels.
# ---------------------------------------------------------------------------
_M64 = (1 << 64) - 1


GUARD_BITS = 5


def guard_round(x64):
    """float64 -> the nearest value with 29 significant bits (a float32 word plus 5 guard
    bits), as float64: add half of mantissa bit 24, clear bits 23..0."""
    b = np.ascontiguousarray(x64, dtype=np.float64).view(np.uint64)
    with np.errstate(over="ignore"):
        b = (b + np.uint64(1 << 23)) & np.uint64(~0xFFFFFF & _M64)
    return b.view(np.float64)


# mode -> (dtype of the float words: prev_shaping / force / x words, container dtype of x, rounding)
STORE_MODES = {"float32": (np.float32, np.float64, "guard"),
               "float32_guard": (np.float32, np.float64, "guard"),
               "float32_rn": (np.float32, np.float32, "rn"),
               "float64": (np.float64, np.float64, "rn")}


# ---------------------------------------------------------------------------
class VecOracle:
    """Batch of independent Lander3D / Hover3D environments on the CPU."""

    def __init__(self, task="lander3d", num_envs=1, tp=TaskParams(), vp=DJI_PHANTOM,
                 substeps=1, store_mode="float64", autoreset=AUTORESET_DISABLED,
                 seed=0, env_id_base=0, time_limit_truncates=False, g=G, mars=None):
        # vp's fields and g may be arrays [n]: per-env vehicles / worlds (domain randomisation);
        # mars = (rho, C_L) (scalars or arrays [n]): the retired Mars model, see refcpu.RigidBody
        assert task in TASKS
        self.task, self.n, self.tp, self.vp = task, int(num_envs), tp, vp
        self.g = g
        self.mars = mars
        self.kind, self.obs_first, self.obs_dim, fan = TASKS[task]
        self.fan = np.array(fan)
        self.act_dim = task_action_dim(task)
        self.substeps = int(substeps)
        word, xdtype, self.rounding = STORE_MODES[store_mode]
        self.T = np.dtype(word)
        self.autoreset = autoreset
        self.seed = int(seed)
        self.env_ids = np.arange(env_id_base, env_id_base + self.n, dtype=np.uint64)
        self.time_limit_truncates = bool(time_limit_truncates)
        self.dt = 1. / (tp.frames_per_second * self.substeps)
        self.max_angle = np.radians(tp.max_angle)
        # episodes started: a plain 32-bit count (the Philox counter word of an episode's draws is episode - 1; 0 =
        # never reset, so 2^32 - 1 is followed by 1).  The step counter below is upstream's (task.py:130): it never
        # saturates.  The device's storage limits (its step counter saturates at 2^S - 1: copterstep_internal.h) are NOT
        # restated here -- a test that runs an env past them applies the documented cap in its comparison
        # (tests/gpu_util.py: device_steps_cap).
        self.episode = np.zeros(self.n, dtype=np.uint32)
        n = self.n
        self.x = np.zeros((12, n), dtype=xdtype)          # struct-of-arrays state
        self.status = np.full(n, LANDED, dtype=np.uint8)
        self.steps = np.zeros(n, dtype=np.int32)
        self.prev_shaping = np.full(n, np.nan, dtype=self.T)   # NaN == "None"
        self.force = np.zeros((3, n), dtype=self.T)       # pending perturbation force [N]
        self.pending = np.zeros(n, dtype=bool)            # perturbation not yet consumed
        self.done_pending = np.zeros(n, dtype=bool)       # NEXT_STEP: reset on next step
        self.ep_return = np.zeros(n)
        self.ep_length = np.zeros(n, dtype=np.int32)
        self.ticks = np.zeros(n, dtype=np.int32)          # Dynamics._ticks (dynamics/__init__.py:98, :197)

    # ------------------------------------------------------------------ physics
    def _physics(self, x, status, pend, k, motors, active):
        """One Dynamics.setMotors() for every lane with active[i]; float64 in, in place.

        x [12,n] f64, status [n] u8, pend [n] bool, k [3,n] f64 (= force / M),
        motors [n,4] f64 (already clipped by the task, or raw in dynamics-only use).
        """
        p = self.vp
        maxrpm = np.asarray(p.maxrpm, dtype=np.float64)
        w = motors * (maxrpm[:, None] if maxrpm.ndim else maxrpm) * np.pi / 30
        w2 = w ** 2
        w0, w1, w2_, w3 = w2[:, 0], w2[:, 1], w2[:, 2], w2[:, 3]
        if self.mars is None:
            U1 = p.B * (((w0 + w1) + w2_) + w3)
            U2 = p.L * p.B * ((w1 + w2_) - (w0 + w3))
            U3 = p.L * p.B * ((w1 + w3) - (w0 + w2_))
            U4 = p.D * ((w0 + w1) - (w2_ + w3))
            Omega = 0
        else:      # attic/mars/dynamics/__init__.py:135-164
            rho, C_L = self.mars
            L = np.asarray(p.L, dtype=np.float64)
            S = .05 * L * 4
            Omega = (w[:, 0] + w[:, 1]) - (w[:, 2] + w[:, 3])
            velocity = w * (L[:, None] if L.ndim else L) / 2
            kl = 0.5 * rho * S * C_L
            lift = (kl[:, None] if np.ndim(kl) else kl) * (velocity ** 2)
            l0, l1, l2, l3 = lift[:, 0], lift[:, 1], lift[:, 2], lift[:, 3]
            U1 = ((l0 + l1) + l2) + l3
            U2 = (l1 + l2) - (l0 + l3)
            U3 = (l1 + l3) - (l0 + l2)
            U4 = p.D * ((w0 + w1) - (w2_ + w3))

        phi, the, psi = x[6], x[8], x[10]
        cph, cth, cps = np.cos(phi), np.cos(the), np.cos(psi)
        sph, sth, sps = np.sin(phi), np.sin(the), np.sin(psi)
        bz = -U1 / p.M
        ax = bz * (sph * sps + cph * cps * sth)
        ay = bz * (cph * sps * sth - cps * sph)
        az = bz * (cph * cth)
        netz = az + self.g

        st = status.copy()
        st[active & (st == LANDED) & (netz < 0)] = AIRBORNE

        leveling = active & (st == LEVELING)
        air = active & (st == AIRBORNE)
        contact = air & (x[4] > 0) & (x[5] > 0)
        hard = (x[5] > LANDING_VEL_Y) | (np.abs(x[3]) > LANDING_VEL_X) | (np.abs(x[6]) > LANDING_ANGLE)
        integ = air & ~contact

        kx = np.where(pend, k[0], 0.0)
        ky = np.where(pend, k[1], 0.0)
        kz = np.where(pend, k[2], 0.0)
        dphi, dthe, dpsi = x[7], x[9], x[11]
        d = np.empty_like(x)
        d[0] = x[1]
        d[1] = ax + kx
        d[2] = x[3]
        d[3] = ay + ky
        d[4] = x[5]
        d[5] = netz + kz
        d[6] = dphi
        d[7] = dpsi * dthe * (p.Iy - p.Iz) / p.Ix - p.Jr / p.Ix * dthe * Omega + U2 / p.Ix + 0.0
        d[8] = dthe
        d[9] = -(dpsi * dphi * (p.Iz - p.Ix) / p.Iy + p.Jr / p.Iy * dphi * Omega + U3 / p.Iy) + 0.0
        d[10] = dpsi
        d[11] = dthe * dphi * (p.Ix - p.Iy) / p.Iz + U4 / p.Iz + 0.0
        d[1] += kx
        d[3] += ky
        d[5] += kz
        d[7] += 0.0
        d[9] += 0.0
        d[11] += 0.0
        xn = x + self.dt * d
        x[:, integ] = xn[:, integ]

        x[6, leveling] = 0
        x[8, leveling] = 0
        st[leveling] = LANDED
        st[contact] = np.where(hard[contact], CRASHED, LEVELING)
        status[:] = st
        # perturbation is consumed, and the clock ticks, in every call that does not freeze on contact
        pend[active & ~contact] = False
        self.ticks[active & ~contact] += 1

    # ------------------------------------------------------------------ reset
    def _reset_lanes(self, m, forces=None, poses=None, perturb=True):
        """Masked reset (task.py:145-197): state, status, perturbation, the
        'initializing' step's shaping, steps = 1.  poses [5, n] = (x, y, altitude, phi_deg,
        theta_deg) per env and perturb: _Task._reset's keywords."""
        tp = self.tp
        if not np.any(m):
            return
        self.x[:, m] = 0
        if poses is None:
            self.x[4, m] = self.T.type(-tp.initial_altitude)
        else:
            p = np.asarray(poses, dtype=np.float64)[:, m]
            x0 = np.zeros((12, p.shape[1]))
            x0[0], x0[2], x0[4] = p[0], p[1], -p[2]
            x0[6], x0[8] = np.radians(p[3]), np.radians(p[4])
            self.x[:, m] = self._round(x0)
        self.status[m] = np.where(self.x[4, m].astype(np.float64) < 0, AIRBORNE, LANDED)
        nxt = self.episode[m].astype(np.int64) + 1
        self.episode[m] = np.where(nxt > 0xFFFFFFFF, 1, nxt).astype(np.uint32)
        if not perturb:
            f = np.zeros((3, int(np.sum(m))))
        elif forces is None:
            f = draw_forces(self.seed, self.env_ids[m], self.episode[m] - np.uint32(1), tp.initial_random_force)
        else:
            f = np.asarray(forces, dtype=np.float64)[:, m]
        self.force[:, m] = f.astype(self.T)
        self.pending[m] = bool(perturb)
        self.done_pending[m] = False
        xs = self.x[:, m].astype(np.float64)
        if self.kind == "lander":
            self.prev_shaping[m] = self._shaping(xs).astype(self.T)
        else:
            self.prev_shaping[m] = np.nan
        self.steps[m] = 1
        self.ticks[m] = 0                                  # a new Dynamics object (task.py:161)
        self.ep_return[m] = 0
        self.ep_length[m] = 0

    def reset(self, mask=None, forces=None, seed=None, poses=None, perturb=True):
        """Reset all lanes (mask None) or lanes with mask[i] != 0.  Returns obs [n, obs_dim] f32."""
        if seed is not None:
            self.seed = int(seed)
        m = np.ones(self.n, dtype=bool) if mask is None else np.asarray(mask).astype(bool)
        self._reset_lanes(m, forces, poses, perturb)
        return self.observe()

    def observe(self):
        # float32 observation = round-to-nearest-even of the stored value, in every mode
        with np.errstate(over="ignore"):
            return np.ascontiguousarray(
                self.x[self.obs_first:self.obs_first + self.obs_dim].T.astype(np.float32))

    # ------------------------------------------------------------------ reward
    def _shaping(self, x):
        tp = self.tp
        s6 = ((((x[0] ** 2 + x[1] ** 2) + x[2] ** 2) + x[3] ** 2) + x[4] ** 2) + x[5] ** 2
        s2 = x[10] ** 2 + x[11] ** 2
        sh = -(tp.xyz_penalty_factor * np.sqrt(s6) + tp.yaw_penalty_factor * np.sqrt(s2))
        return np.where(np.abs(x[5]) > tp.dz_max, sh - tp.dz_penalty, sh)

    # ------------------------------------------------------------------ dynamics-only stepping
    def set_motors(self, motors):
        """Dynamics.setMotors() on every lane with raw (unclipped) motor values: the
        reference's L2 interface, used to replay the D-series traces."""
        x = self.x.astype(np.float64)
        k = self.force.astype(np.float64) / self.vp.M
        motors = np.asarray(motors, dtype=np.float64)
        active = np.ones(self.n, dtype=bool)
        for _ in range(self.substeps):
            self._physics(x, self.status, self.pending, k, motors, active)
        self.x[:] = self._round(x)

    def _round(self, x64):
        """float64 registers -> stored state words."""
        if self.rounding == "guard":
            return guard_round(x64)
        with np.errstate(over="ignore"):
            return x64.astype(self.T)

    # ------------------------------------------------------------------ step
    def step(self, actions):
        tp, n = self.tp, self.n
        actions = np.asarray(actions, dtype=np.float64).reshape(n, self.act_dim)[:, self.fan]   # _get_motors
        obs = np.empty((n, self.obs_dim), dtype=np.float32)
        reward = np.zeros(n)
        term = np.zeros(n, dtype=bool)
        trunc = np.zeros(n, dtype=bool)

        resetting = self.done_pending.copy() if self.autoreset == AUTORESET_NEXT_STEP else np.zeros(n, bool)
        live = ~resetting

        status0 = self.status.copy()
        x = self.x.astype(np.float64)
        k = self.force.astype(np.float64) / self.vp.M
        motors = np.clip(actions, 0, 1)
        active = live & (status0 != LANDED)
        for _ in range(self.substeps):
            self._physics(x, self.status, self.pending, k, motors, active)
        self.x[:, live] = self._round(x)[:, live]

        # task logic on the *stored* (rounded) state, float64 arithmetic
        x = self.x.astype(np.float64)
        done = np.zeros(n, dtype=bool)
        if self.kind == "lander":
            sh = self._shaping(x)
            prev = self.prev_shaping.astype(np.float64)
            r = np.where(np.isnan(prev), 0.0, sh - prev)
            self.prev_shaping[live] = sh.astype(self.T)[live]
            landed0 = status0 == LANDED
            done |= landed0
            inside = np.sqrt(x[0] ** 2 + x[2] ** 2) < tp.target_radius
            r = np.where(landed0 & inside, r + tp.inside_radius_bonus, r)
        else:
            r = np.ones(n)
        oob = (np.abs(x[0]) >= tp.bounds) | (np.abs(x[2]) >= tp.bounds)
        tilt = ~oob & ((np.abs(x[6]) >= self.max_angle) | (np.abs(x[8]) >= self.max_angle))
        crash = ~oob & ~tilt & (status0 == CRASHED)
        r = np.where(oob, r - tp.out_of_bounds_penalty, r)
        r = np.where(tilt, -float(tp.out_of_bounds_penalty), r)
        done |= oob | tilt | crash
        limit = self.steps == tp.max_steps
        if self.time_limit_truncates:
            tr = limit & ~done
        else:
            tr = np.zeros(n, dtype=bool)
            done |= limit
        self.steps[live] = self.steps[live] + 1

        reward[live] = r[live]
        term[live] = done[live]
        trunc[live] = tr[live]
        obs[:] = self.observe()

        fin = live & (term | trunc)
        self.ep_return[live] += r[live]
        self.ep_length[live] += 1
        self.last_done = np.flatnonzero(fin)
        self.last_return = self.ep_return[fin].copy()
        self.last_length = self.ep_length[fin].copy()
        self.final_obs = obs.copy()

        self.last_reset = np.zeros(n, dtype=bool)      # lanes whose episode restarted in this step
        if self.autoreset == AUTORESET_NEXT_STEP:
            self._reset_lanes(resetting)
            obs[resetting] = self.observe()[resetting]
            self.done_pending = fin
            self.last_reset = resetting
        elif self.autoreset == AUTORESET_SAME_STEP:
            self._reset_lanes(fin)
            obs[fin] = self.observe()[fin]
            self.last_reset = fin
        return obs, reward, term, trunc


class VecPid:
    """oracle.refcpu.PidHeuristic over a batch (same float64 operation order, lane by lane):
    attic/mars/pidcontrollers/__init__.py:12-146 + attic/mars/lander3d.py:64-87.  Controller state
    per env: 6 controllers x (errorI, lastError, deltaError1, deltaError2); the last two (yaw rate,
    altitude) belong to the hover heuristic.  `reset(mask)` gives
    the masked envs fresh controllers (a new episode)."""

    ROLL_RATE, PITCH_RATE, ROLL_POS, PITCH_POS, YAW_RATE, ALTITUDE = range(6)

    def __init__(self, num_envs, gains=None):
        from oracle.refcpu import PidGains
        self.g = gains or PidGains()
        self.n = num_envs
        self.state = np.zeros((6, 4, num_envs))        # [controller][errI, last, d1, d2][env]
        self.big = np.radians(self.g.rate_big)

    def reset(self, mask=None):
        if mask is None:
            self.state[:] = 0
        else:
            self.state[:, :, np.asarray(mask).astype(bool)] = 0

    def _compute(self, c, kp, ki, kd, windup, target, actual):
        st = self.state[c]
        error = target - actual
        out = error * kp
        iterm = np.zeros(self.n)
        if ki > 0:
            st[0] = np.clip(st[0] + error, -windup, windup)
            iterm = st[0] * ki
        out = out + iterm
        dterm = np.zeros(self.n)
        if kd > 0:
            de = error - st[1]
            dterm = ((st[2] + st[3]) + de) * kd
            st[3] = st[2].copy()
            st[2] = de
            st[1] = error
        return out + dterm

    def _rate(self, c, w):
        g = self.g
        wild = np.abs(w) > self.big
        self.state[c][0][wild] = 0
        self.state[c][1][wild] = 0
        return self._compute(c, g.rate_kp, g.rate_ki, g.rate_kd, g.rate_windup, 0.0, w)

    def _pos(self, c, x, dx):
        g = self.g
        return self._compute(c, g.pos_kp, g.pos_ki, g.pos_kd, g.pos_windup, (g.pos_target - x) * 1, dx)

    def action(self, obs):
        """obs [n, >=10] float32 -> [n, 4] float32 motor demands (unclipped)."""
        o = np.asarray(obs, dtype=np.float32).astype(np.float64).T
        x, dx, y, dy, z, dz, phi, dphi, theta, dtheta = o[:10]
        r = self._rate(self.ROLL_RATE, dphi) + self._pos(self.ROLL_POS, y, dy)
        p = self._rate(self.PITCH_RATE, -dtheta) + self._pos(self.PITCH_POS, x, dx)
        g = self.g
        if g.heuristic == "hover":                 # attic/mars/hover3d.py:65-92 (needs the 12-slot observation)
            yw = self._rate(self.YAW_RATE, -o[11])
            hover = self._compute(self.ALTITUDE, g.alt_kp, g.alt_ki, g.alt_kd, g.alt_windup,
                                  (g.alt_target - (-z)) * 1, -dz)
            t = (hover + 1) / 2
            a = np.stack([t - r - p - yw, t + r + p - yw, t + r - p + yw, t - r + p + yw], axis=1)
        else:
            t = ((z * g.descent_kp + dz * g.descent_kd) + 1) / 2
            a = np.stack([t - r - p, t + r + p, t + r - p, t - r + p], axis=1)
        with np.errstate(over="ignore"):
            return a.astype(np.float32)
