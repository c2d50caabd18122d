"""CopterVecEnv: Gymnasium-style vector environment whose step()/reset() are one HIP
kernel launch each, through the C ABI of libcopterstep.so.

It mirrors the interface the reference exposes through gym.make('gym_copter:Lander-v0')
(reference gym_copter/envs/task.py:23-143, envs/lander.py:15-97, gym_copter/__init__.py:9-13),
widened from one environment to a batch:

    reference (single env)                        here (batch of N)
    ---------------------------------------------------------------------------------
    reset(seed, options) -> (obs[10], {})         reset(seed, options) -> (obs[N,10], {})
    step(a[4]) -> (obs, r, done, False, {})       step(a[N,4]) -> (obs[N,10], r[N], term[N], trunc[N], {})
    observation_space / action_space              single_*_space + batched *_space
    set_altitude(a), close(), unwrapped, FRAMES_PER_SECOND, STATE_NAMES, metadata

Device tensors go in and come out zero-copy (torch CUDA/HIP tensors); NumPy actions are
accepted for drop-in use and then NumPy arrays are returned (paying PCIe both ways).
PyTorch is used only for device memory and streams.  There is no CPU implementation in
this package: construction fails if libcopterstep.so or a HIP device is missing.
"""
import atexit
import ctypes as C
import os
import weakref

import numpy as np

from . import _lib
from .spaces import gymnasium_api

# Envs that hold a device context.  Whatever is still open when the interpreter exits is closed by an exit handler,
# i.e. BEFORE modules and the HIP runtime's own exit handlers are torn down: a context destroyed from __del__ during
# interpreter shutdown calls into a runtime that may already be half gone.
_open_envs = weakref.WeakSet()
_owner_pid = os.getpid()


def _close_open_envs():
    if os.getpid() != _owner_pid:      # a forked child: the contexts are the parent's
        return
    for env in list(_open_envs):
        try:
            env.close()
        except Exception:
            pass


atexit.register(_close_open_envs)

# With Gymnasium importable CopterVecEnv IS a gymnasium.vector.VectorEnv with gymnasium.spaces.Box spaces and a
# gymnasium.vector.AutoresetMode in its metadata (what gymnasium.make_vec and VectorEnv consumers check); without it
# (gymnasium is not a dependency) the same attributes on a plain class (spaces.py).
_VectorEnvBase, Box, batch_space, _AUTORESET_META, HAVE_GYMNASIUM = gymnasium_api()

_TASKS = {"lander3d": _lib.TASK_LANDER3D, "lander": _lib.TASK_LANDER3D,
          "hover3d": _lib.TASK_HOVER3D, "hover": _lib.TASK_HOVER3D,
          # 2D / 1D variants (attic lander2d.py / lander1d.py / hover2d.py / hover1d.py hooks)
          "lander2d": _lib.TASK_LANDER2D, "lander1d": _lib.TASK_LANDER1D,
          "hover2d": _lib.TASK_HOVER2D, "hover1d": _lib.TASK_HOVER1D}
_TASK_NAMES = {_lib.TASK_LANDER3D: "lander3d", _lib.TASK_HOVER3D: "hover3d", _lib.TASK_LANDER2D: "lander2d",
               _lib.TASK_LANDER1D: "lander1d", _lib.TASK_HOVER2D: "hover2d", _lib.TASK_HOVER1D: "hover1d"}
# task -> (first observed state slot, observation size, action size)
_TASK_SHAPES = {"lander3d": (0, 10, 4), "hover3d": (0, 12, 4), "lander2d": (2, 6, 2), "hover2d": (2, 6, 2),
                "lander1d": (4, 2, 1), "hover1d": (4, 2, 1)}
# float32 (default) = float32 state words + 5 guard bits; see DESIGN.md "state words"
_STATE_MODES = {"float32": _lib.STATE_F32G, "float32_guard": _lib.STATE_F32G,
                "float32_rn": _lib.STATE_F32_RN, "float64": _lib.STATE_F64}
_AUTORESET = {"disabled": _lib.AUTORESET_DISABLED, "next_step": _lib.AUTORESET_NEXT_STEP,
              "same_step": _lib.AUTORESET_SAME_STEP}
_VEHICLE_KEYS = ("B", "D", "M", "L", "Ix", "Iy", "Iz", "Jr", "maxrpm")   # dji_phantom.py:9-26
# how the motor model is evaluated: float64 (Python-float / float64 actions upstream) or NumPy's
# float32 path for float32 action arrays (dynamics/__init__.py:120-132 under NumPy >= 2 promotion)
_ARITH = {"float64": _lib.ARITH_F64, "float32": _lib.ARITH_F32}
# thrust law: live B*omega^2, or the retired Mars model's lift-coefficient law
_THRUST = {"B": _lib.THRUST_B, "lift": _lib.THRUST_LIFT}
_TASK_KEYS = {"initial_random_force": "initial_random_force",             # task.py:32-38
              "out_of_bounds_penalty": "out_of_bounds_penalty",
              "max_angle": "max_angle_deg", "bounds": "bounds",
              "initial_altitude": "initial_altitude",
              # Lander's class constants (lander.py:17-23), overridable upstream by subclassing
              "target_radius": "target_radius", "yaw_penalty_factor": "yaw_penalty_factor",
              "xyz_penalty_factor": "xyz_penalty_factor", "dz_max": "dz_max", "dz_penalty": "dz_penalty",
              "inside_radius_bonus": "inside_radius_bonus"}

# default outputs: packed rows up to this many envs (one wavefront per SIMD on 256 CUs x 2), plain arrays above
PACKED_ROWS_MAX_ENVS = 131072

STATE_NAMES_12 = ['X', 'dX', 'Y', 'dY', 'Z', 'dZ', 'Phi', 'dPhi', 'Theta', 'dTheta', 'Psi', 'dPsi']


def _torch():
    import torch
    return torch


class CopterVecEnv(_VectorEnvBase):
    FRAMES_PER_SECOND = 100                                    # task.py:25
    # class-level defaults of the gymnasium.vector.VectorEnv attribute set (instances overwrite them)
    metadata = {"render_modes": [], "render_fps": 100}         # task.py:27-30 (rendering is out of scope: no modes)
    render_mode = None
    spec = None                                                # gymnasium.make_vec assigns env.unwrapped.spec
    closed = False

    def __init__(self, task="lander3d", num_envs=1, device=0, seed=0,
                 autoreset_mode="next_step", substeps=1, state_dtype="float32",
                 time_limit_truncates=False, episode_stats=False, env_id_base=0,
                 max_steps=1000, vehicle_params=None, frames_per_second=None,
                 action_arith="float64", thrust_model="B", rotor_gyro=False, world_params=None,
                 track_time=False, copy=True, contiguous_outputs=False, **task_kwargs):
        # the constructor's keywords as given: what pickling reproduces (__reduce__ below), as the reference's
        # EzPickle does for its envs (task.py:23, :40)
        self._ctor_kwargs = dict(task=task, num_envs=num_envs, device=device, seed=seed, autoreset_mode=autoreset_mode,
                                 substeps=substeps, state_dtype=state_dtype, time_limit_truncates=time_limit_truncates,
                                 episode_stats=episode_stats, env_id_base=env_id_base, max_steps=max_steps,
                                 vehicle_params=None if vehicle_params is None else dict(vehicle_params),
                                 frames_per_second=frames_per_second, action_arith=action_arith,
                                 thrust_model=thrust_model, rotor_gyro=rotor_gyro,
                                 world_params=None if world_params is None else dict(world_params),
                                 track_time=track_time, copy=copy, contiguous_outputs=contiguous_outputs,
                                 **task_kwargs)
        lib = _lib.load()
        torch = _torch()
        if task not in _TASKS:
            raise ValueError("unknown task %r (have %s)" % (task, sorted(_TASKS)))
        if state_dtype not in _STATE_MODES:
            raise ValueError("unknown state_dtype %r (have %s)" % (state_dtype, sorted(_STATE_MODES)))
        if autoreset_mode not in _AUTORESET:
            raise ValueError("unknown autoreset_mode %r (have %s)" % (autoreset_mode, sorted(_AUTORESET)))
        if isinstance(device, str):
            device = torch.device(device).index or 0
        elif isinstance(device, torch.device):
            device = device.index or 0
        self._lib = lib
        self._ctx = C.c_void_p()
        cfg = _lib.Config()
        _lib.check(lib.cs_config_init(C.byref(cfg), _TASKS[task]))
        cfg.state_mode = _STATE_MODES[state_dtype]
        cfg.autoreset = _AUTORESET[autoreset_mode]
        cfg.substeps = int(substeps)
        cfg.time_limit_truncates = int(bool(time_limit_truncates))
        cfg.episode_stats = int(bool(episode_stats))
        cfg.device = int(device)
        cfg.max_steps = int(max_steps)
        cfg.num_envs = int(num_envs)
        cfg.env_id_base = int(env_id_base)
        cfg.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        if frames_per_second is not None:
            cfg.frames_per_second = float(frames_per_second)
            self.FRAMES_PER_SECOND = frames_per_second
        for k, v in (vehicle_params or {}).items():
            if k not in _VEHICLE_KEYS + ("C_L",):
                raise ValueError("unknown vehicle parameter %r" % k)
            setattr(cfg, k, float(v))
        for k, v in (world_params or {}).items():      # attic/mars/dynamics/__init__.py:85-86
            if k not in ("G", "rho"):
                raise ValueError("unknown world parameter %r" % k)
            setattr(cfg, k, float(v))
        if action_arith not in _ARITH:
            raise ValueError("action_arith must be one of %s" % sorted(_ARITH))
        if thrust_model not in _THRUST:
            raise ValueError("thrust_model must be one of %s" % sorted(_THRUST))
        cfg.action_arith = _ARITH[action_arith]
        cfg.thrust_model = _THRUST[thrust_model]
        cfg.rotor_gyro = int(bool(rotor_gyro))
        cfg.track_time = int(bool(track_time))      # Dynamics._ticks / getTime(), dynamics/__init__.py:197, :219-221
        for k, v in task_kwargs.items():
            if k not in _TASK_KEYS:
                raise TypeError("unexpected keyword argument %r" % k)
            setattr(cfg, _TASK_KEYS[k], float(v))
        self.config = cfg
        self.task = _TASK_NAMES[cfg.task]
        self.num_envs = int(num_envs)
        self.autoreset_mode = autoreset_mode
        self.episode_stats = bool(episode_stats)
        self.track_time = bool(track_time)
        self.copy = bool(copy)              # as gymnasium.vector.SyncVectorEnv(copy=True): NumPy returns are the caller's
        # Default outputs (see _open_device): up to PACKED_ROWS_MAX_ENVS envs step() / reset() return STRIDED views --
        # the columns of one [n, obs_dim + 2] array (obs has row stride obs_dim + 2: obs.view(-1) raises, use
        # .reshape / .contiguous()).  contiguous_outputs=True allocates four plain contiguous arrays at every size
        # instead (+0.5 ... 2 % per step below 131 072 envs; what gather="obs" sharding and obs.view(...) callers want).
        self.contiguous_outputs = bool(contiguous_outputs)
        self.device = torch.device("cuda", int(device))
        first, self.obs_dim, self.action_dim = _TASK_SHAPES[self.task]
        self.STATE_NAMES = STATE_NAMES_12[first:first + self.obs_dim]   # lander.py:30-31
        # metadata["autoreset_mode"]: a gymnasium.vector.AutoresetMode member (NEXT_STEP / SAME_STEP / DISABLED) when
        # Gymnasium is importable -- make_vec warns about anything else -- and a look-alike with the same name and
        # value otherwise; the plain string stays in self.autoreset_mode
        self.metadata = {"render_modes": [], "render_fps": self.FRAMES_PER_SECOND,
                         "autoreset_mode": _AUTORESET_META[autoreset_mode]}
        self.render_mode = None
        self.spec = None
        self.single_observation_space = Box(-np.inf, np.inf, (self.obs_dim,), np.float32)  # task.py:46-49
        self.single_action_space = Box(-1, +1, (self.action_dim,), np.float32)              # task.py:52-55
        self.observation_space = batch_space(self.single_observation_space, self.num_envs)
        self.action_space = batch_space(self.single_action_space, self.num_envs)
        self.closed = False
        self._fast = self._final_obs = self._done = None
        self._open_device()

    def _open_device(self):
        """The device half of construction: the context (cs_create) and the default output buffers."""
        torch = _torch()
        lib, cfg = self._lib, self.config
        # cs_create fails loudly when no HIP device is usable (no CPU fallback)
        _lib.check(lib.cs_create(C.byref(cfg), C.byref(self._ctx)))
        _open_envs.add(self)
        od, ad = C.c_int32(), C.c_int32()
        _lib.check(lib.cs_obs_dim(self._ctx, C.byref(od)))
        _lib.check(lib.cs_action_dim(self._ctx, C.byref(ad)))
        assert (od.value, ad.value) == (self.obs_dim, self.action_dim), "library / binding disagree on shapes"
        n = self.num_envs
        with torch.cuda.device(self.device):
            # the default outputs.  Up to PACKED_ROWS_MAX_ENVS envs: the columns of ONE [n, obs_dim + 2] float32 array --
            # "packed rows" (include/copterstep.h, cs_step_io): row i = {observation, reward, flags word}; the step kernel
            # writes whole rows (one output stream per wavefront instead of three: -0.5 ... -2 % per step while a SIMD
            # holds one wavefront).  Above: plain obs / reward arrays + the two flags as the columns of one [n,2] byte
            # array (interleaved flags) -- at >= 262 144 envs the 2 extra bytes per env of a packed row and its ragged
            # last store cost 0.6 ... 3 % (round 4).  Either way step() returns views, and the NumPy
            # convenience path ships one array to the host.
            from .sharded import row_views
            self._flags2 = None
            if self.contiguous_outputs:
                self._rows = None
                self._obs = torch.zeros((n, self.obs_dim), dtype=torch.float32, device=self.device)
                self._reward = torch.zeros(n, dtype=torch.float32, device=self.device)
                self._term = torch.zeros(n, dtype=torch.uint8, device=self.device)
                self._trunc = torch.zeros(n, dtype=torch.uint8, device=self.device)
            elif 1 < n <= int(os.environ.get("COPTERSTEP_PACKED_ROWS_MAX_ENVS", PACKED_ROWS_MAX_ENVS)):
                self._rows = torch.zeros((n, self.obs_dim + 2), dtype=torch.float32, device=self.device)
                self._obs, self._reward, self._term, self._trunc = row_views(self._rows, self.obs_dim)
            else:
                self._rows = None
                fl = self._flags2 = torch.zeros((n, 2), dtype=torch.uint8, device=self.device)
                self._obs = torch.zeros((n, self.obs_dim), dtype=torch.float32, device=self.device)
                self._reward = torch.zeros(n, dtype=torch.float32, device=self.device)
                self._term, self._trunc = fl[:, 0], fl[:, 1]
            self._obs_plain = None          # contiguous [n, obs_dim] scratch for entry points that write plain rows
            self._serve_out = None
            self._final_obs = None
            self._done = None
        self._cache_outputs()

    def bind_outputs(self, obs, reward, terminated, truncated):
        """Make step()/reset() write into caller-provided device tensors (same shapes and dtypes
        as the defaults; truncated/terminated as uint8): four contiguous arrays, or the flags as the columns of
        one [N,2] tensor, or all four as the columns of one [N, obs_dim + 2] float32 array (packed rows, e.g.
        gym_copter_amd.sharded.PackedOutputs: what a single collective then ships)."""
        torch = _torch()
        n, od = self.num_envs, self.obs_dim
        for t, shape, dt in ((obs, (n, od), torch.float32), (reward, (n,), torch.float32),
                             (terminated, (n,), torch.uint8), (truncated, (n,), torch.uint8)):
            if tuple(t.shape) != shape or t.dtype != dt or t.device != self.device:
                raise ValueError("bind_outputs: need %s %s on %s" % (dt, shape, self.device))
        # three accepted forms (include/copterstep.h, cs_step_io): (a) four contiguous arrays; (b) the flags as the two
        # columns of one [n,2] uint8 array, the rest contiguous; (c) all four the columns of ONE [n, obs_dim + 2]
        # float32 array (packed rows)
        base = obs.data_ptr()
        packed = (n > 1 and obs.stride() == (od + 2, 1) and reward.stride() == (od + 2,)
                  and reward.data_ptr() == base + 4 * od and terminated.stride() == (4 * (od + 2),)
                  and terminated.data_ptr() == base + 4 * (od + 1) and truncated.stride() == (4 * (od + 2),)
                  and truncated.data_ptr() == terminated.data_ptr() + 1)
        interleaved = (n > 1 and terminated.stride() == (2,) and truncated.stride() == (2,)
                       and truncated.data_ptr() == terminated.data_ptr() + 1)
        if not packed:
            if not (obs.is_contiguous() and reward.is_contiguous()):
                raise ValueError("bind_outputs: obs and reward must be contiguous (or all four outputs the columns of one "
                                 "(%d, %d) float32 array)" % (n, od + 2))
            if not (interleaved or (terminated.is_contiguous() and truncated.is_contiguous())):
                raise ValueError("bind_outputs: terminated / truncated must be contiguous uint8 (%d,) tensors, or the two "
                                 "columns of one (%d, 2) uint8 tensor" % (n, n))
        self._rows = self._flags2 = None     # (the default arrays are no longer what step() writes)
        self._obs, self._reward, self._term, self._trunc = obs, reward, terminated, truncated
        self._cache_outputs()

    def _cache_outputs(self):
        """The output buffers are persistent: their device pointers and the bool views of the flag
        buffers are computed once, not per step (the eager step path is host-bound)."""
        torch = _torch()
        self._out_ptrs = tuple(C.c_void_p(t.data_ptr())
                               for t in (self._obs, self._reward, self._term, self._trunc))
        # cs_step_io.output_form (ABI 5): this wrapper knows what it allocated / was bound to, and says so where it
        # passes a cs_step_io; the bare-pointer cs_step infers the same (packed rows only for num_envs > 1, and
        # bind_outputs / the default allocation never use them for one env)
        n, od = self.num_envs, self.obs_dim
        self._output_form = (_lib.OUTPUT_PACKED_ROWS if (n > 1 and self._reward.data_ptr() == self._obs.data_ptr() + 4 * od
                                                        and self._obs.stride(0) == od + 2) else _lib.OUTPUT_PLAIN)
        self._term_b, self._trunc_b = self._term.view(torch.bool), self._trunc.view(torch.bool)
        self._dev_index = self.device.index
        # raw current-stream / current-device queries (no Stream object, no lazy-init check); fall back
        # to the public API
        self._raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
        self._cur_device = getattr(torch._C, "_cuda_getDevice", torch.cuda.current_device)
        # the per-step call: cs_step by address through the _cs_call module when it is built (same entry
        # point, no ctypes marshalling), else through ctypes
        self._Tensor, self._f32 = torch.Tensor, torch.float32
        self._ashape = (self.num_envs, self.action_dim)
        self._fast = None
        try:
            from . import _cs_call
            addr = C.cast(self._lib.cs_step, C.c_void_p).value
            if self._raw_stream is not None and addr:
                self._fast = (_cs_call.step, addr, self._ctx.value) + tuple(p.value for p in self._out_ptrs)
        except ImportError:
            pass

    # -- plumbing ------------------------------------------------------------------
    @property
    def unwrapped(self):
        return self

    def __reduce__(self):
        """Pickling = the constructor keywords, exactly what the reference's envs pickle through
        gymnasium.utils.EzPickle (task.py:23, :40: `EzPickle.__init__(self)` records the constructor arguments and
        unpickling calls the constructor again).  The copy is a FRESH env on the same device index -- an env factory
        for multiprocessing evaluators (attic/neat/README.md:21-23); simulation state does not travel (use
        get_state() / set_state() for a checkpoint).  Nothing touches the device until the copy is built."""
        kw = dict(self._ctor_kwargs)
        dev = kw.get("device")
        if not isinstance(dev, (int, str)):            # a torch.device: keep what identifies it
            kw["device"] = getattr(dev, "index", None) or 0
        return (_rebuild_env, (kw,))

    def _stream(self):
        if self._raw_stream is not None:
            return C.c_void_p(self._raw_stream(self._dev_index))
        return C.c_void_p(_torch().cuda.current_stream(self.device).cuda_stream)

    def _dev_f32(self, a, shape, name):
        """Return (device float32 contiguous tensor view/copy, was_numpy)."""
        torch = _torch()
        if not isinstance(a, (torch.Tensor, np.ndarray)):
            # array-likes of other libraries (the reference takes whatever np.clip takes, task.py:91): a device array
            # that speaks DLPack or __cuda_array_interface__ (CuPy, JAX, Numba, ...) is adopted in place -- no host
            # round trip; anything else goes through NumPy below
            if hasattr(a, "__dlpack__") and hasattr(a, "__dlpack_device__"):
                a = torch.from_dlpack(a)
            elif hasattr(a, "__cuda_array_interface__"):
                a = torch.as_tensor(a, device=self.device)
            elif type(a).__name__ == "PyCapsule":       # a bare DLPack capsule
                a = torch.utils.dlpack.from_dlpack(a)
            if isinstance(a, torch.Tensor) and a.device.type == "cpu":
                a = a.numpy()                           # a host array under another name: the NumPy path
        was_numpy = not isinstance(a, torch.Tensor)
        if was_numpy:
            arr = np.asarray(a, dtype=np.float32)
            if name == "actions" and arr.shape == tuple(shape):
                # NumPy actions every step: stage them in pinned memory (an H2D from pageable memory is a
                # synchronous double copy) and keep a resident destination
                st = getattr(self, "_act_stage", None)
                if st is None:
                    st = self._act_stage = (torch.empty(shape, dtype=torch.float32).pin_memory(),
                                            torch.empty(shape, dtype=torch.float32, device=self.device),
                                            torch.cuda.Event())
                else:
                    st[2].synchronize()            # the previous upload has left the pinned buffer
                st[0].numpy()[...] = arr
                st[1].copy_(st[0], non_blocking=True)
                st[2].record(torch.cuda.current_stream(self.device))
                return st[1], True
            t = torch.from_numpy(np.ascontiguousarray(arr))
        else:
            t = a
        if tuple(t.shape) != tuple(shape):
            raise ValueError("%s must have shape %s, got %s" % (name, tuple(shape), tuple(t.shape)))
        if t.device != self.device or t.dtype != torch.float32 or not t.is_contiguous():
            t = t.to(device=self.device, dtype=torch.float32, non_blocking=True).contiguous()
        return t, was_numpy

    def _check_open(self):
        if self.closed:
            raise RuntimeError("environment is closed")

    # -- Gymnasium surface ---------------------------------------------------------
    def seed(self, seed=None):                                  # task.py:71-75
        self._check_open()
        _lib.check(self._lib.cs_seed(self._ctx, int(seed or 0) & 0xFFFFFFFFFFFFFFFF))
        return [seed]

    def set_altitude(self, altitude):                           # task.py:67-69
        self._check_open()
        _lib.check(self._lib.cs_set_altitude(self._ctx, float(altitude)))
        self.config.initial_altitude = float(altitude)     # reset(options={'perturb': False}) starts from it

    def reset(self, seed=None, options=None):
        """Reset every env (or options['mask']); returns (obs[N,obs_dim], {}).

        options: {'mask': bool[N], 'forces': float[3,N] newtons (else Philox U[-F,F)),
        'pose': float[5,N] or [5] = (x, y, altitude, roll_deg, pitch_deg) and 'perturb': bool --
        _Task._reset's keywords (task.py:145)}.
        seed re-keys the perturbation stream (the reference draws from global np.random,
        task.py:199-202; here the draw is counter-based on (seed, global env id, episode #))."""
        self._check_open()
        torch = _torch()
        options = options or {}
        if seed is not None:
            self.seed(seed)
        mask = options.get("mask")
        forces = options.get("forces")
        mask_t = force_t = None
        mask_p = force_p = None
        if mask is not None:
            mask_t = torch.as_tensor(np.asarray(mask) if not isinstance(mask, torch.Tensor) else mask)
            mask_t = (mask_t != 0).to(device=self.device, dtype=torch.uint8).contiguous()
            if tuple(mask_t.shape) != (self.num_envs,):
                raise ValueError("mask must have shape (%d,)" % self.num_envs)
            mask_p = C.c_void_p(mask_t.data_ptr())
        if forces is not None:
            force_t, _ = self._dev_f32(forces, (3, self.num_envs), "forces")
            force_p = C.c_void_p(force_t.data_ptr())
        pose, perturb = options.get("pose"), bool(options.get("perturb", True))
        pose_t = None
        if pose is not None or not perturb:
            if pose is None:
                pose = (0.0, 0.0, float(self.config.initial_altitude), 0.0, 0.0)
            if not isinstance(pose, torch.Tensor):
                pose = np.asarray(pose, dtype=np.float32)
                if pose.shape == (5,):
                    pose = np.repeat(pose[:, None], self.num_envs, axis=1)
            pose_t, _ = self._dev_f32(pose, (5, self.num_envs), "pose")
        # cs_reset writes plain [N, obs_dim] rows: straight into the observation buffer when that is contiguous, else
        # (packed rows) into a scratch buffer that is then copied into the observation columns (resets are rare)
        plain = self._obs
        if not plain.is_contiguous():
            if self._obs_plain is None:
                self._obs_plain = torch.empty((self.num_envs, self.obs_dim), dtype=torch.float32, device=self.device)
            plain = self._obs_plain
        with torch.cuda.device(self.device):
            if pose_t is None:
                _lib.check(self._lib.cs_reset(self._ctx, mask_p, force_p,
                                              C.c_void_p(plain.data_ptr()), self._stream()))
            else:
                _lib.check(self._lib.cs_reset_pose(self._ctx, mask_p, C.c_void_p(pose_t.data_ptr()), int(perturb),
                                                   force_p, C.c_void_p(plain.data_ptr()), self._stream()))
            if plain is not self._obs:
                self._obs.copy_(plain)
        self._keep = (mask_t, force_t, pose_t)      # alive until the stream has consumed them
        if (forces is not None and perturb and self.config.state_mode == _lib.STATE_F64
                and not isinstance(forces, torch.Tensor)):
            # float64 state words: the device entry point takes float32 force rows; install the float64 values
            # the caller gave (upstream's force / M is float64) through the host path where they differ
            f64 = np.ascontiguousarray(np.asarray(forces, dtype=np.float64))
            if not np.array_equal(f64, f64.astype(np.float32).astype(np.float64)):
                st = self.get_state(only=("force", "flags"))
                m = np.ones(self.num_envs, bool) if mask is None else (np.asarray(to_numpy_mask(mask)) != 0)
                st["force"][:, m] = f64[:, m]
                self.set_state(force=st["force"], flags=(st["flags"] | np.where(m, 5, 0)).astype(np.uint8))
        return self._obs, {}

    def step(self, actions):
        """One env step for the whole batch: exactly one kernel launch, asynchronous on
        the current torch stream.  Returned tensors are this env's persistent output
        buffers (overwritten by the next step())."""
        fast = self._fast
        if (fast is not None and type(actions) is self._Tensor and actions.dtype is self._f32
                and actions.shape == self._ashape and actions.device == self.device and actions.is_contiguous()
                and self._final_obs is None and self._done is None and not self.closed
                and self._cur_device() == self._dev_index):
            # eager fast path: a resident float32 action batch, default outputs, this env's device current
            rc = fast[0](fast[1], fast[2], actions.data_ptr(), fast[3], fast[4], fast[5], fast[6],
                         self._raw_stream(self._dev_index))
            if rc != 0:
                _lib.check(rc)
            self._keep = actions
            return self._obs, self._reward, self._term_b, self._trunc_b, {}
        self._check_open()
        torch = _torch()
        a, was_numpy = self._dev_f32(actions, (self.num_envs, self.action_dim), "actions")
        if self._final_obs is None and self._done is None and self._cur_device() == self._dev_index:
            # resident actions, default outputs, the env's device already current
            po, pr, pt, pu = self._out_ptrs
            rc = self._lib.cs_step(self._ctx, C.c_void_p(a.data_ptr()), po, pr, pt, pu, self._stream())
            if rc != 0:
                _lib.check(rc)
            self._keep = a
            if was_numpy:
                return self._outputs_to_numpy() + ({},)
            return self._obs, self._reward, self._term_b, self._trunc_b, {}
        with torch.cuda.device(self.device):
            if self._final_obs is None and self._done is None:
                _lib.check(self._lib.cs_step(
                    self._ctx, C.c_void_p(a.data_ptr()), C.c_void_p(self._obs.data_ptr()),
                    C.c_void_p(self._reward.data_ptr()), C.c_void_p(self._term.data_ptr()),
                    C.c_void_p(self._trunc.data_ptr()), self._stream()))
            else:
                io = _lib.StepIO()
                io.output_form = self._output_form
                io.actions_dev = a.data_ptr()
                io.obs_dev = self._obs.data_ptr()
                io.reward_dev = self._reward.data_ptr()
                io.terminated_dev = self._term.data_ptr()
                io.truncated_dev = self._trunc.data_ptr()
                if self._final_obs is not None:
                    io.final_obs_dev = self._final_obs.data_ptr()
                if self._done is not None:
                    io.done_count_dev = self._done["count"].data_ptr()
                    io.done_ids_dev = self._done["ids"].data_ptr()
                    io.done_length_dev = self._done["length"].data_ptr()
                    if self.episode_stats:
                        io.done_return_dev = self._done["return"].data_ptr()
                _lib.check(self._lib.cs_step_ex(self._ctx, C.byref(io), self._stream()))
        self._keep = a
        infos = {}
        if self._final_obs is not None:
            infos["final_obs"] = self._final_obs
        if self._done is not None:
            infos["episode"] = self._done
        term, trunc = self._term_b, self._trunc_b
        if was_numpy:
            return self._outputs_to_numpy() + ({k: _to_numpy(v) for k, v in infos.items()},)
        return self._obs, self._reward, term, trunc, infos

    def _outputs_to_numpy(self):
        """The NumPy convenience path.  Packed rows (the default up to PACKED_ROWS_MAX_ENVS envs): obs, reward and both
        flags cross PCIe as ONE device-to-host copy and the arrays returned are views of that host array (obs has row
        stride obs_dim + 2).  Plain default arrays (larger batches, contiguous_outputs=True): one copy per array
        straight from the buffers the kernel wrote -- obs, reward and the [n,2] flags array (or the two flag arrays);
        no device-side repacking.  Caller-bound outputs of any other shape are gathered into packed rows first.
        copy=True (the default, as gymnasium.vector.SyncVectorEnv): fresh host buffers every step -- the arrays are
        the caller's to keep.  copy=False: two sets of PINNED buffers alternate (no staging copy, no allocation), so
        what a step returned stays valid only until the step after the next one."""
        torch = _torch()
        n, od = self.num_envs, self.obs_dim
        rows = self._rows
        plain = None
        if rows is None:
            if self._obs.is_contiguous() and self._reward.is_contiguous():
                if self._flags2 is not None:
                    plain = (self._obs, self._reward, self._flags2)
                elif self._term.is_contiguous() and self._trunc.is_contiguous():
                    plain = (self._obs, self._reward, self._term, self._trunc)
        if plain is not None:
            def host_like(t):
                return torch.empty(t.shape, dtype=t.dtype)
            if self.copy:
                hosts = [host_like(t) for t in plain]
            else:
                sets = getattr(self, "_plain_host", None)
                if sets is None or len(sets[0]) != len(plain):
                    sets = self._plain_host = [[host_like(t).pin_memory() for t in plain] for _ in (0, 1)]
                    self._pack_turn = 0
                hosts = sets[self._pack_turn]
                self._pack_turn ^= 1
            for h, t in zip(hosts[:-1], plain[:-1]):
                h.copy_(t, non_blocking=not self.copy)      # (pinned destinations: asynchronous, ordered on the stream)
            hosts[-1].copy_(plain[-1])                      # the last one blocks: all of them have landed
            if not self.copy:
                torch.cuda.current_stream(self.device).synchronize()
            hn = [h.numpy() for h in hosts]
            if len(hn) == 3:
                fb = hn[2].view(np.bool_)
                return hn[0], hn[1], fb[:, 0], fb[:, 1]
            return hn[0], hn[1], hn[2].view(np.bool_), hn[3].view(np.bool_)
        if rows is None:                    # outputs re-bound by the caller in another shape: gather them into packed rows
            from .sharded import row_views
            rows = getattr(self, "_rows_tmp", None)
            if rows is None:
                rows = self._rows_tmp = torch.zeros((n, od + 2), dtype=torch.float32, device=self.device)
            o, r, t, u = row_views(rows, od)
            o.copy_(self._obs)
            r.copy_(self._reward)
            t.copy_(self._term)
            u.copy_(self._trunc)
        if self.copy:
            host = torch.empty((n, od + 2), dtype=torch.float32)
        else:
            hosts = getattr(self, "_pack_host", None)
            if hosts is None:
                hosts = self._pack_host = [torch.empty((n, od + 2), dtype=torch.float32).pin_memory() for _ in (0, 1)]
                self._pack_turn = 0
            host = hosts[self._pack_turn]
            self._pack_turn ^= 1
        host.copy_(rows)                                   # the one blocking D2H
        h = host.numpy()                                   # (shares the tensor's memory and keeps it alive)
        hb = h.view(np.bool_)                              # [n, 4 * (od + 2)]: the flag bytes are 0 / 1
        return h[:, :od], h[:, od], hb[:, 4 * (od + 1)], hb[:, 4 * (od + 1) + 1]

    def step_many(self, actions):
        """K steps in ONE kernel launch for resident action batches: actions [K,N,4] ->
        (obs [K,N,obs_dim], reward [K,N], terminated [K,N], truncated [K,N]).  Bit-identical to
        K calls of step(actions[k]); the env state stays in registers between the steps."""
        self._check_open()
        torch = _torch()
        if not isinstance(actions, torch.Tensor):
            actions = torch.from_numpy(np.ascontiguousarray(np.asarray(actions, dtype=np.float32)))
        if actions.dim() != 3 or tuple(actions.shape[1:]) != (self.num_envs, self.action_dim):
            raise ValueError("actions must have shape (K, %d, %d), got %s"
                             % (self.num_envs, self.action_dim, tuple(actions.shape)))
        a = actions.to(device=self.device, dtype=torch.float32).contiguous()
        K, n = int(a.shape[0]), self.num_envs
        buf = getattr(self, "_many", None)
        if buf is None or buf[0].shape[0] != K:
            flags = torch.empty((K, n, 2), dtype=torch.uint8, device=self.device)     # interleaved flags
            buf = (torch.empty((K, n, self.obs_dim), dtype=torch.float32, device=self.device),
                   torch.empty((K, n), dtype=torch.float32, device=self.device), flags[:, :, 0], flags[:, :, 1])
            self._many = buf
        p = lambda t: C.c_void_p(t.data_ptr())
        with torch.cuda.device(self.device):
            _lib.check(self._lib.cs_step_many(self._ctx, K, p(a), p(buf[0]), p(buf[1]), p(buf[2]),
                                              p(buf[3]), self._stream()))
        self._keep = a
        return buf[0], buf[1], buf[2].view(torch.bool), buf[3].view(torch.bool)

    # -- per-env vehicles / worlds (domain randomisation) --------------------------------
    VEHICLE_ROWS = _VEHICLE_KEYS + ("G", "rho", "C_L")

    def set_vehicle_params(self, params=None, **columns):
        """Give every env its own vehicle and world: `params` is [12, N] (rows B, D, M, L, Ix,
        Iy, Iz, Jr, maxrpm -- the reference's `vehicle_params` keys, dji_phantom.py:9-26 -- then G,
        rho, C_L: Dynamics.G and the Mars model's air density and lift coefficient), or pass columns
        by name (scalars or [N]); unnamed ones keep this env's configured values.
        set_vehicle_params(None) returns to the uniform vehicle."""
        self._check_open()
        torch = _torch()
        if params is None and not columns:
            with torch.cuda.device(self.device):
                _lib.check(self._lib.cs_set_vehicle_params(self._ctx, None))
            return None
        n = self.num_envs
        if params is None:
            base = [getattr(self.config, k) for k in self.VEHICLE_ROWS]
            table = np.repeat(np.asarray(base, dtype=np.float64)[:, None], n, axis=1)
            for k, v in columns.items():
                if k not in self.VEHICLE_ROWS:
                    raise TypeError("unknown vehicle parameter %r (have %s)" % (k, self.VEHICLE_ROWS))
                table[self.VEHICLE_ROWS.index(k)] = np.asarray(v, dtype=np.float64)
        else:
            table = np.asarray(params, dtype=np.float64).reshape(-1, n)
            if table.shape[0] == 10:      # vehicle + G only: the air of this env's configuration
                air = np.repeat(np.array([[self.config.rho], [self.config.C_L]]), n, axis=1)
                table = np.concatenate([table, air], axis=0)
            if table.shape[0] != len(self.VEHICLE_ROWS):
                raise ValueError("params must have %d rows %s (or the first 10), got %d"
                                 % (len(self.VEHICLE_ROWS), self.VEHICLE_ROWS, table.shape[0]))
        table = np.ascontiguousarray(table)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.cs_set_vehicle_params(self._ctx, table.ctypes.data_as(C.c_void_p)))
        return table

    # -- closed-loop rollouts under the on-device PID landing heuristic ----------------
    def configure_pid(self, heuristic="lander", **gains):
        """Install a PID heuristic with the controllers of attic/mars/pidcontrollers:
        heuristic="lander" = attic/mars/lander3d.py:32-36, :64-87 (descent law); "hover" =
        attic/mars/hover3d.py:65-92 (yaw-rate + altitude-hold controllers; Hover3D only).  Keywords
        override upstream's gains: rate_kp, rate_ki, rate_kd, rate_windup, rate_big_deg, pos_kp,
        pos_ki, pos_kd, pos_target, pos_windup, descent_kp, descent_kd, alt_kp, alt_ki, alt_kd,
        alt_target, alt_windup.  Returns the gains in effect."""
        self._check_open()
        g = _lib.PidGains()
        _lib.check(self._lib.cs_pid_gains_init(C.byref(g)))
        if heuristic not in ("lander", "hover"):
            raise ValueError("heuristic must be 'lander' or 'hover'")
        g.heuristic = _lib.PID_HOVER if heuristic == "hover" else _lib.PID_LANDER
        for k, v in gains.items():
            if k in ("struct_size", "heuristic") or not hasattr(g, k):
                raise TypeError("unknown PID gain %r" % (k,))
            setattr(g, k, float(v))
        torch = _torch()
        with torch.cuda.device(self.device):
            _lib.check(self._lib.cs_pid_configure(self._ctx, C.byref(g)))
        self._pid = True
        return {k: getattr(g, k) for k, _ in g._fields_[2:]}

    def rollout_random(self, num_steps, return_actions=False):
        """K steps in ONE kernel launch under the on-device random policy (actions ~ U[-1,1)
        drawn in the kernel from Philox keyed by seed, global env id, episode and step): the
        `action_space.sample()` loop with no action tensor.  Returns like rollout_pid."""
        return self._rollout(self._lib.cs_rollout_random, num_steps, return_actions)

    def rollout_pid(self, num_steps, return_actions=False):
        """K closed-loop steps in ONE kernel launch: every step's action is the PID heuristic of
        the observation the previous step returned.  -> (obs [K,N,obs_dim], reward [K,N],
        terminated [K,N], truncated [K,N]) and, with return_actions, the float32 actions [K,N,4]
        appended."""
        if not getattr(self, "_pid", False):
            self.configure_pid()
        return self._rollout(self._lib.cs_rollout_pid, num_steps, return_actions)

    def rollout_policy(self, policy, num_steps, params=None, return_actions=False):
        """K closed-loop steps in ONE kernel launch under the CALLER'S OWN policy: `policy` is a HIP device
        functor compiled by gym_copter_amd.compile_policy(env, source) and fused into the K-step kernel
        (include/copterstep_rollout.h); `params` is the float32 device tensor its first member points at (weights,
        per-env policy state, ...; None for a policy without parameters).  Returns like rollout_pid."""
        torch = _torch()
        if (policy.task, policy.state_mode) != (self.task, int(self.config.state_mode)):
            raise ValueError("this policy was compiled for task %r / storage mode %d" % (policy.task, policy.state_mode))
        if params is not None:
            if not isinstance(params, torch.Tensor):
                params = torch.as_tensor(np.asarray(params, dtype=np.float32))
            params = params.to(device=self.device, dtype=torch.float32).contiguous()
        self._policy_keep = (policy, params)                 # alive until the stream has consumed them
        pp = C.c_void_p(params.data_ptr()) if params is not None else None
        entry = lambda ctx, K, a, o, r, t, u, stream: policy._entry(ctx, K, pp, a, o, r, t, u, stream)
        return self._rollout(entry, num_steps, return_actions)

    def _rollout(self, entry, num_steps, return_actions):
        self._check_open()
        torch = _torch()
        K, n = int(num_steps), self.num_envs
        buf = getattr(self, "_roll", None)
        if buf is None or buf[0].shape[0] != K:
            flags = torch.empty((K, n, 2), dtype=torch.uint8, device=self.device)     # interleaved flags
            buf = (torch.empty((K, n, self.obs_dim), dtype=torch.float32, device=self.device),
                   torch.empty((K, n), dtype=torch.float32, device=self.device), flags[:, :, 0], flags[:, :, 1],
                   torch.empty((K, n, self.action_dim), dtype=torch.float32, device=self.device))
            self._roll = buf
        p = lambda t: C.c_void_p(t.data_ptr())
        with torch.cuda.device(self.device):
            _lib.check(entry(self._ctx, K, p(buf[4]) if return_actions else None,
                             p(buf[0]), p(buf[1]), p(buf[2]), p(buf[3]), self._stream()))
        out = (buf[0], buf[1], buf[2].view(torch.bool), buf[3].view(torch.bool))
        return out + (buf[4],) if return_actions else out

    # -- served stepping: one persistent env kernel per session (cs_serve_*) -------------------
    def serve_max_envs(self):
        """Largest batch a served session accepts on this device (every tile's wavefront stays resident)."""
        out = C.c_int64()
        _lib.check(self._lib.cs_serve_max_envs(self._ctx, C.byref(out)))
        return out.value

    def serve_begin(self, num_steps, ring=4, timeout=2.0):
        """Open a served session of `num_steps` steps: ONE persistent kernel keeps every env in registers
        and takes each step's action rows from, and publishes its outputs to, tagged granule rings in
        device memory -- the policy <-> step() loop (reference lander.py:40-65) without a kernel launch
        per env step.  Feed it with serve_submit / serve_collect (plain tensors), serve_policy_pid (a
        policy kernel per step) or your own HIP kernels (include/copterstep_serve.h); close with
        serve_end().  Enqueued on the current stream.  serve_begin / serve_end are eager calls; the feeder
        launches between them may be captured into a graph once and replayed against every later session.
        Returns the wire description (a _lib.ServeView)."""
        self._check_open()
        torch = _torch()
        view = _lib.ServeView()
        with torch.cuda.device(self.device):
            _lib.check(self._lib.cs_serve_begin(self._ctx, int(num_steps), int(ring), float(timeout),
                                                self._stream(), C.byref(view)))
        return view

    def serve_submit(self, step, actions):
        """Plain action rows [N,A] (device float32) of step `step` -> the session's action ring."""
        a, _ = self._dev_f32(actions, (self.num_envs, self.action_dim), "actions")
        with _torch().cuda.device(self.device):
            _lib.check(self._lib.cs_serve_submit(self._ctx, int(step), C.c_void_p(a.data_ptr()), self._stream()))
        self._keep = a

    def serve_collect(self, step, out=None):
        """Wait (on the device) for the outputs of step `step` (-1: the observation before step 0) and
        return them as step() would: (obs, reward, terminated, truncated), by default in this env's
        persistent output buffers, or in `out` = (obs, reward, terminated u8, truncated u8) tensors."""
        torch = _torch()
        if out is None and self._reward.data_ptr() == self._obs.data_ptr() + 4 * self.obs_dim:     # packed rows
            # (cs_serve_collect writes plain arrays, not packed rows: its own contiguous buffers)
            if self._serve_out is None:
                n, dev = self.num_envs, self.device
                fl = torch.zeros((n, 2), dtype=torch.uint8, device=dev)
                self._serve_out = (torch.empty((n, self.obs_dim), dtype=torch.float32, device=dev),
                                   torch.empty(n, dtype=torch.float32, device=dev), fl[:, 0], fl[:, 1])
            out = self._serve_out
        obs, rew, term, trunc = out if out is not None else (self._obs, self._reward, self._term, self._trunc)
        p = lambda t: C.c_void_p(t.data_ptr())
        with torch.cuda.device(self.device):
            _lib.check(self._lib.cs_serve_collect(self._ctx, int(step), p(obs), p(rew), p(term), p(trunc),
                                                  self._stream()))
        return obs, rew, term.view(torch.bool), trunc.view(torch.bool)

    def serve_policy_pid(self, step, num_steps=1):
        """One closed-loop policy step as its own kernel: the PID heuristic of configure_pid() on the
        outputs of step - 1 -> the actions of `step`.  num_steps > 1: the policy of the steps [step, step +
        num_steps) as ONE persistent kernel next to the env kernel (controllers in registers, no launch in
        the loop)."""
        if not getattr(self, "_pid", False):
            self.configure_pid()
        with _torch().cuda.device(self.device):
            _lib.check(self._lib.cs_serve_policy_pid_many(self._ctx, int(step), int(num_steps), self._stream()))

    def serve_end(self, wait=True):
        """Close the session: order the current stream behind the env kernel's exit.  wait=True also waits for
        it -> steps every tile completed; raises CopterStepError(code ERR_TIMEOUT) if a wavefront gave up
        waiting for its actions.  wait=False only enqueues (-> None; serve_status() reports later)."""
        done = C.c_int32(-1)
        with _torch().cuda.device(self.device):
            _lib.check(self._lib.cs_serve_end(self._ctx, self._stream(), C.byref(done) if wait else None))
        return done.value if wait else None

    def serve_status(self):
        """(steps completed by every tile, by the fastest tile, wavefronts that gave up) of the last
        session; synchronises the env kernel's stream."""
        lo, hi, to = C.c_int32(), C.c_int32(), C.c_int32()
        rc = self._lib.cs_serve_status(self._ctx, C.byref(lo), C.byref(hi), C.byref(to))
        if rc not in (0, _lib.ERR_TIMEOUT):
            _lib.check(rc)
        return lo.value, hi.value, to.value

    def pid_get_state(self):
        """Controller state as a host array [24, N] float64 (rows: see include/copterstep.h)."""
        self._check_open()
        out = np.empty((_lib.PID_ROWS, self.num_envs), dtype=np.float64)
        torch = _torch()
        with torch.cuda.device(self.device):
            _lib.check(self._lib.cs_pid_get_state(self._ctx, out.ctypes.data_as(C.c_void_p), self._stream()))
        return out

    def pid_set_state(self, state):
        self._check_open()
        st = np.ascontiguousarray(np.asarray(state, dtype=np.float64).reshape(_lib.PID_ROWS, self.num_envs))
        torch = _torch()
        with torch.cuda.device(self.device):
            _lib.check(self._lib.cs_pid_set_state(self._ctx, st.ctypes.data_as(C.c_void_p), self._stream()))

    def clock_probe(self, waves_per_simd=4):
        """The shader clock (Hz) this device holds under a float64 vector load (cs_clock_probe: one ~0.3 ms kernel,
        synchronous): what the instruction-issue bounds of the K-step kernels are priced at by bench.py."""
        self._check_open()
        hz = C.c_double()
        with _torch().cuda.device(self.device):
            _lib.check(self._lib.cs_clock_probe(self._ctx, int(waves_per_simd), C.byref(hz), self._stream()))
        return hz.value

    def pci_address(self):
        """'dddd:bb:dd.f' of this env's device (cs_device_pci_address): its sysfs node is /sys/bus/pci/devices/<it>."""
        self._check_open()
        buf = C.create_string_buffer(32)
        _lib.check(self._lib.cs_device_pci_address(self._ctx, buf, 32))
        return buf.value.decode()

    def close(self, **kwargs):                                  # task.py:139-143 (gymnasium.vector.VectorEnv.close(**kwargs))
        if not self.closed and self._ctx:
            self._lib.cs_destroy(self._ctx)
            self._ctx = C.c_void_p()
        self.closed = True
        _open_envs.discard(self)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- optional outputs ------------------------------------------------------------
    def enable_final_obs(self):
        """SAME_STEP autoreset: also return the pre-reset observation in infos['final_obs']."""
        torch = _torch()
        self._final_obs = torch.zeros((self.num_envs, self.obs_dim), dtype=torch.float32,
                                      device=self.device)

    def enable_done_list(self):
        """Compacted list of finished envs per step (wave-ballot compaction on device):
        infos['episode'] = {'count': i32[1], 'ids': i32[N], 'length': i32[N], 'return': f32[N]}."""
        torch = _torch()
        n, dev = self.num_envs, self.device
        self._done = {"count": torch.zeros(1, dtype=torch.int32, device=dev),
                      "ids": torch.zeros(n, dtype=torch.int32, device=dev),
                      "length": torch.zeros(n, dtype=torch.int32, device=dev)}
        if self.episode_stats:
            self._done["return"] = torch.zeros(n, dtype=torch.float32, device=dev)

    # -- Dynamics-level access (reference dynamics/__init__.py public methods) ---------
    def set_motors(self, motors):
        """`substeps` x Dynamics.setMotors(motors[i]) on every env, no task logic."""
        self._check_open()
        torch = _torch()
        m, _ = self._dev_f32(motors, (self.num_envs, 4), "motors")
        with torch.cuda.device(self.device):
            _lib.check(self._lib.cs_set_motors(self._ctx, C.c_void_p(m.data_ptr()), self._stream()))
        self._keep = m

    def state_tensors(self):
        """Dynamics.getState() / getStatus() for the batch as DEVICE tensors, asynchronous on the
        current stream: {'x': float32 [12, N] (upstream slot order, incl. psi / dpsi), 'status':
        uint8 [N], 'steps': int32 [N], 'ticks': int32 [N] (Dynamics._ticks; -1 without track_time)}.  The
        tensors are persistent buffers of this env."""
        self._check_open()
        torch = _torch()
        if getattr(self, "_state_t", None) is None:
            n = self.num_envs
            self._state_t = {"x": torch.empty((12, n), dtype=torch.float32, device=self.device),
                             "status": torch.empty(n, dtype=torch.uint8, device=self.device),
                             "steps": torch.empty(n, dtype=torch.int32, device=self.device),
                             "ticks": torch.empty(n, dtype=torch.int32, device=self.device)}
        t = self._state_t
        with torch.cuda.device(self.device):
            _lib.check(self._lib.cs_export_state(self._ctx, C.c_void_p(t["x"].data_ptr()),
                                                 C.c_void_p(t["status"].data_ptr()),
                                                 C.c_void_p(t["steps"].data_ptr()),
                                                 C.c_void_p(t["ticks"].data_ptr()), self._stream()))
        return t

    def get_time(self):
        """Dynamics.getTime() (dynamics/__init__.py:219-221) for the batch: ticks * dt as a float64 device
        tensor [N]; needs track_time=True."""
        if not self.track_time:
            raise RuntimeError("get_time() needs CopterVecEnv(track_time=True)")
        dt = 1.0 / (float(self.config.frames_per_second) * int(self.config.substeps))
        return self.state_tensors()["ticks"].double() * dt

    def get_state(self, only=None):
        """Whole-batch state as NumPy (synchronises): dict with x[12,N] f64, status, steps,
        prev_shaping (NaN = None), force[3,N] newtons (this episode's reset perturbation: an installed one, or
        the Philox draw of (seed, global env id, episode - 1)), flags (bit 0 perturbation pending, bit 1 reset
        pending, bit 2 the perturbation was installed explicitly), episode, (episode_return), (ticks).
        set_state(**get_state()) is a faithful restore: a `force` that comes with `flags` is installed only
        where bit 2 says it was explicit; the other envs stay on their Philox draw.  only=("x", ...) fetches
        just those arrays."""
        self._check_open()
        n = self.num_envs
        out = {"x": np.empty((12, n)), "status": np.empty(n, np.uint8), "steps": np.empty(n, np.int32),
               "prev_shaping": np.empty(n), "force": np.empty((3, n)), "flags": np.empty(n, np.uint8),
               "episode": np.empty(n, np.uint32)}
        er = np.empty(n) if self.episode_stats else None
        tk = np.empty(n, np.int32) if self.track_time else None
        if only is not None:             # only these arrays cross PCIe (the others are not even staged)
            out = {k: v for k, v in out.items() if k in only}
            er = er if "episode_return" in only else None
            tk = tk if "ticks" in only else None
        p = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
        g = out.get
        _lib.check(self._lib.cs_get_state(self._ctx, p(g("x")), p(g("status")), p(g("steps")),
                                          p(g("prev_shaping")), p(g("force")), p(g("flags")),
                                          p(er), p(g("episode")), p(tk), self._stream()))
        if er is not None:
            out["episode_return"] = er
        if tk is not None:
            out["ticks"] = tk
        return out

    def set_state(self, x=None, status=None, steps=None, prev_shaping=None, force=None, flags=None,
                  episode_return=None, episode=None, ticks=None):
        self._check_open()
        n = self.num_envs

        def prep(a, shape, dtype):
            if a is None:
                return None
            a = np.ascontiguousarray(np.asarray(a, dtype=dtype))
            if a.shape != shape:
                raise ValueError("expected shape %s, got %s" % (shape, a.shape))
            return a
        arrs = [prep(x, (12, n), np.float64), prep(status, (n,), np.uint8), prep(steps, (n,), np.int32),
                prep(prev_shaping, (n,), np.float64), prep(force, (3, n), np.float64),
                prep(flags, (n,), np.uint8), prep(episode_return, (n,), np.float64),
                prep(episode, (n,), np.uint32), prep(ticks, (n,), np.int32)]
        ptrs = [None if a is None else a.ctypes.data_as(C.c_void_p) for a in arrs]
        _lib.check(self._lib.cs_set_state(self._ctx, *ptrs, self._stream()))


    def set_perturbation(self, force_xyz, mask=None):
        """Dynamics.perturb() (dynamics/__init__.py:227-229) for the batch (or the envs of `mask`):
        install a pending force [3,N] in newtons that the next integrating physics call consumes
        (applied twice in that call, as upstream does).  One kernel launch on the current stream."""
        self._check_open()
        torch = _torch()
        if self.config.state_mode == _lib.STATE_F64 and mask is None and not isinstance(force_xyz, torch.Tensor):
            # float64 state words: keep the force in float64 (upstream's force / M is float64); the device
            # entry point takes float32 rows, the host one float64
            f64 = np.ascontiguousarray(np.asarray(force_xyz, dtype=np.float64))
            if f64.shape != (3, self.num_envs):
                raise ValueError("force_xyz must have shape (3, %d)" % self.num_envs)
            flags = self.get_state()["flags"]
            self.set_state(force=f64, flags=(flags | 1 | 4).astype(np.uint8))
            return
        f, _ = self._dev_f32(force_xyz, (3, self.num_envs), "force_xyz")
        mask_t, mask_p = None, None
        if mask is not None:
            mask_t = torch.as_tensor(np.asarray(mask) if not isinstance(mask, torch.Tensor) else mask)
            mask_t = (mask_t != 0).to(device=self.device, dtype=torch.uint8).contiguous()
            if tuple(mask_t.shape) != (self.num_envs,):
                raise ValueError("mask must have shape (%d,)" % self.num_envs)
            mask_p = C.c_void_p(mask_t.data_ptr())
        with torch.cuda.device(self.device):
            _lib.check(self._lib.cs_set_perturbation(self._ctx, mask_p, C.c_void_p(f.data_ptr()), self._stream()))
        self._keep = (f, mask_t)

    perturb = set_perturbation

    STATS_NAMES = ("envs", "airborne", "steps_sum", "steps_max", "episodes_started", "return_sum", "nonfinite")

    def batch_stats(self):
        """Batch bookkeeping reduced on the device (cs_episode_stats): a float64 tensor [7] =
        (envs, envs airborne, sum and max of the episode step counters, episodes started, sum of the
        running episode returns, envs with a NaN / inf state word -- the guard counter for what upstream
        lets propagate silently, task.py:133), asynchronous on the current stream."""
        self._check_open()
        torch = _torch()
        if getattr(self, "_stats_t", None) is None:
            self._stats_t = torch.zeros(_lib.EPISODE_STATS, dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.cs_episode_stats(self._ctx, C.c_void_p(self._stats_t.data_ptr()), self._stream()))
        return self._stats_t

    def set_tuning(self, nt_action_max_envs=0, nt_state_min_envs=0, direct_rows_max_envs=0):
        """Launcher thresholds (0 = built-in default; they pick between instantiations of the same
        kernel and never change results).  Returns the values in effect."""
        self._check_open()
        t = _lib.Tuning(C.sizeof(_lib.Tuning), int(nt_action_max_envs), int(nt_state_min_envs),
                        int(direct_rows_max_envs))
        _lib.check(self._lib.cs_set_tuning(self._ctx, C.byref(t)))
        return self.get_tuning()

    def get_tuning(self):
        t = _lib.Tuning()
        _lib.check(self._lib.cs_get_tuning(self._ctx, C.byref(t)))
        return {"nt_action_max_envs": t.nt_action_max_envs, "nt_state_min_envs": t.nt_state_min_envs,
                "direct_rows_max_envs": t.direct_rows_max_envs}


def _rebuild_env(kwargs):
    return CopterVecEnv(**kwargs)


def to_numpy_mask(mask):
    return mask.detach().cpu().numpy() if hasattr(mask, "detach") else mask


def _to_numpy(v):
    if isinstance(v, dict):
        return {k: _to_numpy(x) for k, x in v.items()}
    return v.cpu().numpy()
