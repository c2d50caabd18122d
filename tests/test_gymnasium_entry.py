"""The Gymnasium entry point, executed (missing #1).

The reference is reached through gym.make('gym_copter:Lander-v0') (reference gym_copter/__init__.py:9-13,
lander.py:80).  Gymnasium is absent from the build image, so these CPU tests install an in-memory stand-in for the
part of Gymnasium >= 1.0 that registration and make_vec touch -- registry / EnvSpec / register(vector_entry_point=...) /
make_vec with its VECTOR_ENTRY_POINT branch (entry(num_envs=N, max_episode_steps=..., **kwargs), env.unwrapped.spec
assignment, the metadata["autoreset_mode"] check), gymnasium.vector.VectorEnv / AutoresetMode, gymnasium.spaces.Box,
gymnasium.vector.utils.batch_space -- re-import gym_copter_amd against it, and run the path once.  The device half of
CopterVecEnv's constructor (cs_create + output buffers) is replaced by a no-op: no GPU is needed, everything else is
the product's own code.
"""
import copy
import enum
import importlib
import logging
import re
import sys
import types
import warnings
from dataclasses import dataclass, field

import numpy as np
import pytest


# ------------------------------------------------------------------------------------------------
# the stand-in (behaviour restated from Gymnasium 1.x's envs/registration.py and vector/vector_env.py)
# ------------------------------------------------------------------------------------------------
def build_gymnasium_standin():
    gym = types.ModuleType("gymnasium")

    class Box:
        def __init__(self, low, high, shape=None, dtype=np.float32, seed=None):
            self.shape = tuple(shape) if shape is not None else np.shape(low)
            self.dtype = np.dtype(dtype)
            self.low = np.full(self.shape, low, dtype=self.dtype)
            self.high = np.full(self.shape, high, dtype=self.dtype)

        def sample(self):
            return np.zeros(self.shape, self.dtype)

        def contains(self, x):
            return np.shape(x) == self.shape

        def __eq__(self, other):      # (gymnasium.spaces.Box.__eq__: same type, shape, dtype and bounds)
            return (isinstance(other, Box) and self.shape == other.shape and self.dtype == other.dtype
                    and np.array_equal(self.low, other.low) and np.array_equal(self.high, other.high))

    class AutoresetMode(enum.Enum):
        NEXT_STEP = "NextStep"
        SAME_STEP = "SameStep"
        DISABLED = "Disabled"

    class VectorEnv:
        metadata = {}
        spec = None
        render_mode = None
        closed = False
        observation_space = action_space = single_observation_space = single_action_space = None
        num_envs = None
        _np_random = None
        _np_random_seed = None

        def reset(self, *, seed=None, options=None):
            raise NotImplementedError

        def step(self, actions):
            raise NotImplementedError

        def close(self, **kwargs):
            if self.closed:
                return
            self.close_extras(**kwargs)
            self.closed = True

        def close_extras(self, **kwargs):
            pass

        @property
        def unwrapped(self):
            return self

        def __del__(self):
            if not getattr(self, "closed", True):
                self.close()

    def batch_space(space, n=1):
        return Box(space.low.flat[0], space.high.flat[0], shape=(n,) + tuple(space.shape), dtype=space.dtype)

    class VectorizeMode(enum.Enum):
        ASYNC = "async"
        SYNC = "sync"
        VECTOR_ENTRY_POINT = "vector_entry_point"

    @dataclass
    class EnvSpec:
        id: str
        entry_point: object = None
        reward_threshold: object = None
        nondeterministic: bool = False
        max_episode_steps: object = None
        order_enforce: bool = True
        disable_env_checker: bool = False
        kwargs: dict = field(default_factory=dict)
        additional_wrappers: tuple = ()
        vector_entry_point: object = None

    class Error(Exception):
        pass

    registry = {}
    ENV_ID_RE = re.compile(r"^(?:(?P<namespace>[\w:-]+)\/)?(?:(?P<name>[\w:.-]+?))(?:-v(?P<version>\d+))?$")

    def register(id, entry_point=None, reward_threshold=None, nondeterministic=False, max_episode_steps=None,
                 order_enforce=True, disable_env_checker=False, additional_wrappers=(), vector_entry_point=None,
                 kwargs=None):
        assert entry_point is not None or vector_entry_point is not None, "Either `entry_point` or `vector_entry_point` (or both) must be provided"
        if ENV_ID_RE.fullmatch(id) is None:
            raise Error("Malformed environment ID: %s" % id)
        if id in registry:
            warnings.warn("Overriding environment %s already in registry." % id)
        registry[id] = EnvSpec(id=id, entry_point=entry_point, reward_threshold=reward_threshold,
                               nondeterministic=nondeterministic, max_episode_steps=max_episode_steps,
                               order_enforce=order_enforce, disable_env_checker=disable_env_checker,
                               kwargs=dict(kwargs or {}), additional_wrappers=tuple(additional_wrappers),
                               vector_entry_point=vector_entry_point)

    def load_env_creator(name):
        mod_name, attr_name = name.split(":")
        return getattr(importlib.import_module(mod_name), attr_name)

    def make_vec(id, num_envs=1, vectorization_mode=None, vector_kwargs=None, wrappers=None, **kwargs):
        vector_kwargs = {} if vector_kwargs is None else vector_kwargs
        wrappers = [] if wrappers is None else wrappers
        if id not in registry:
            raise Error("No registered env with id: %s" % id)
        env_spec = copy.deepcopy(registry[id])
        env_spec_kwargs = env_spec.kwargs
        env_spec.kwargs = dict()
        num_envs = env_spec_kwargs.pop("num_envs", num_envs)
        vectorization_mode = env_spec_kwargs.pop("vectorization_mode", vectorization_mode)
        env_spec_kwargs.update(kwargs)
        if vectorization_mode is None:
            vectorization_mode = (VectorizeMode.VECTOR_ENTRY_POINT if env_spec.vector_entry_point is not None
                                  else VectorizeMode.SYNC)
        assert vectorization_mode == VectorizeMode.VECTOR_ENTRY_POINT, "the stand-in models the vector entry point only"
        if len(vector_kwargs) > 0 or len(wrappers) > 0 or len(env_spec.additional_wrappers) > 0:
            raise Error("Custom vector environment can be passed arguments only through kwargs")
        entry_point = env_spec.vector_entry_point
        env_creator = entry_point if callable(entry_point) else load_env_creator(entry_point)
        if env_spec.max_episode_steps is not None and "max_episode_steps" not in env_spec_kwargs:
            env_spec_kwargs["max_episode_steps"] = env_spec.max_episode_steps
        env = env_creator(num_envs=num_envs, **env_spec_kwargs)
        copied = copy.deepcopy(env_spec)
        copied.kwargs = env_spec_kwargs.copy()
        if num_envs != 1:
            copied.kwargs["num_envs"] = num_envs
        copied.kwargs["vectorization_mode"] = vectorization_mode.value
        env.unwrapped.spec = copied
        if "autoreset_mode" not in env.metadata:
            warnings.warn("The VectorEnv (%s) is missing AutoresetMode metadata, metadata=%s" % (env, env.metadata))
        elif not isinstance(env.metadata["autoreset_mode"], AutoresetMode):
            warnings.warn("The VectorEnv (%s) metadata['autoreset_mode'] is not an instance of AutoresetMode, %s"
                          % (env, type(env.metadata["autoreset_mode"])))
        return env

    spaces = types.ModuleType("gymnasium.spaces")
    spaces.Box = Box
    vector = types.ModuleType("gymnasium.vector")
    vector.VectorEnv, vector.AutoresetMode = VectorEnv, AutoresetMode
    vutils = types.ModuleType("gymnasium.vector.utils")
    vutils.batch_space = batch_space
    vector.utils = vutils
    envs = types.ModuleType("gymnasium.envs")
    registration = types.ModuleType("gymnasium.envs.registration")
    registration.register, registration.registry, registration.make_vec = register, registry, make_vec
    registration.EnvSpec, registration.VectorizeMode = EnvSpec, VectorizeMode
    envs.registration = registration
    error = types.ModuleType("gymnasium.error")
    error.Error = Error
    gym.spaces, gym.vector, gym.envs, gym.error = spaces, vector, envs, error
    gym.register, gym.make_vec, gym.registry = register, make_vec, registry
    gym.__version__ = "1.1.0-standin"
    return {"gymnasium": gym, "gymnasium.spaces": spaces, "gymnasium.vector": vector,
            "gymnasium.vector.utils": vutils, "gymnasium.envs": envs, "gymnasium.envs.registration": registration,
            "gymnasium.error": error}


def _drop(prefixes):
    saved = {}
    for name in list(sys.modules):
        if any(name == p or name.startswith(p + ".") for p in prefixes):
            saved[name] = sys.modules.pop(name)
    return saved


def real_gymnasium():
    """The installed Gymnasium (>= 1.0: make_vec with vector_entry_point), or None.  The build image has none; the
    first user WITH the package runs these same tests against the real thing instead of the stand-in."""
    try:
        import gymnasium
        from gymnasium.envs.registration import register, registry  # noqa: F401
        from gymnasium.vector import VectorEnv  # noqa: F401
        return gymnasium if hasattr(gymnasium, "make_vec") else None
    except Exception:
        return None


@pytest.fixture
def gca_with_gymnasium():
    """gym_copter_amd re-imported against Gymnasium -- the REAL package when it is importable, else the stand-in; the
    original modules are put back afterwards.  -> (gym_copter_amd, gymnasium, constructors seen, is_stand_in)."""
    real = real_gymnasium()
    saved = _drop(("gym_copter_amd", "gym-copter_amd") + (() if real else ("gymnasium",)))
    if real is None:
        mods = build_gymnasium_standin()
        sys.modules.update(mods)
    gym = real if real is not None else mods["gymnasium"]
    try:
        gca = importlib.import_module("gym_copter_amd")
        opened = []
        gca.CopterVecEnv._open_device = lambda self: opened.append(self)      # the device half: not here
        yield gca, gym, opened
    finally:
        if real is not None:      # leave the real registry as it was found
            for full in list(getattr(gca, "_GYMNASIUM_IDS", [])):
                real.envs.registration.registry.pop(full, None)
        _drop(("gym_copter_amd",) + (() if real else ("gymnasium",)))
        sys.modules.update(saved)


def test_ids_are_registered_as_the_reference_registers_lander_v0(gca_with_gymnasium):
    gca, gym, _ = gca_with_gymnasium
    reg = gym.envs.registration.registry
    want = ["gym_copter_amd/%s" % k for k in ("Lander-v0", "Lander3D-v0", "Hover3D-v0", "Lander2D-v0", "Lander1D-v0",
                                               "Hover2D-v0", "Hover1D-v0")]
    assert sorted(gca._GYMNASIUM_IDS) == sorted(want)
    for full in want:
        spec = reg[full]
        assert spec.max_episode_steps == 1000                       # reference gym_copter/__init__.py:12
        assert spec.vector_entry_point == "gym_copter_amd:_vector_entry_point"
        assert spec.kwargs == {"copter_id": full.split("/")[1]}
    # a second import-time call finds them registered and does not register twice (no "Overriding" warning)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        assert sorted(gca.register_with_gymnasium()) == sorted(want)


def test_make_vec_reaches_the_batch_env_with_the_vector_env_attribute_set(gca_with_gymnasium):
    gca, gym, opened = gca_with_gymnasium
    with warnings.catch_warnings():
        warnings.simplefilter("error")          # make_vec warns about a missing / mistyped metadata["autoreset_mode"]
        env = gym.make_vec("gym_copter_amd/Lander-v0", num_envs=7, seed=3)
    assert opened == [env]                                            # the constructor ran up to its device half
    assert type(env) is gca.CopterVecEnv and isinstance(env, gym.vector.VectorEnv)
    assert gca.vecenv.HAVE_GYMNASIUM
    assert env.num_envs == 7 and env.task == "lander3d" and env.config.num_envs == 7 and env.config.seed == 3
    assert env.config.max_steps == 1000                               # max_episode_steps -> the env's own limit
    assert env.unwrapped is env and env.spec is not None and env.spec.id == "gym_copter_amd/Lander-v0"
    assert env.spec.kwargs["num_envs"] == 7
    assert str(getattr(env.spec.kwargs["vectorization_mode"], "value", env.spec.kwargs["vectorization_mode"])) == "vector_entry_point"
    assert env.metadata["autoreset_mode"] is gym.vector.AutoresetMode.NEXT_STEP
    assert env.metadata["render_fps"] == 100 and env.metadata["render_modes"] == []       # task.py:27-30
    assert env.render_mode is None and env.closed is False
    box = gym.spaces.Box
    assert isinstance(env.single_observation_space, box) and env.single_observation_space.shape == (10,)
    assert isinstance(env.single_action_space, box) and env.single_action_space.shape == (4,)
    assert isinstance(env.observation_space, box) and env.observation_space.shape == (7, 10)
    assert isinstance(env.action_space, box) and env.action_space.shape == (7, 4)
    assert gca.Box is box and env.single_action_space == gca.Box(-1, 1, (4,), np.float32)     # the class actually in use
    assert env.single_action_space.low.min() == -1 and env.single_action_space.high.max() == 1       # task.py:52-55
    assert np.isinf(env.single_observation_space.high).all() and env.single_observation_space.dtype == np.float32
    assert env.FRAMES_PER_SECOND == 100 and env.STATE_NAMES[0] == "X" and len(env.STATE_NAMES) == 10
    env.close()
    assert env.closed is True
    with pytest.raises(RuntimeError):
        env.step(np.zeros((7, 4), np.float32))


def test_make_vec_keywords_reach_the_constructor(gca_with_gymnasium):
    gca, gym, _ = gca_with_gymnasium
    env = gym.make_vec("gym_copter_amd/Hover3D-v0", num_envs=3, max_episode_steps=50, autoreset_mode="same_step",
                       initial_altitude=4.0)
    assert env.task == "hover3d" and env.config.max_steps == 50 and env.autoreset_mode == "same_step"
    assert env.metadata["autoreset_mode"] is gym.vector.AutoresetMode.SAME_STEP
    assert env.config.initial_altitude == 4.0 and env.observation_space.shape == (3, 12)
    env2 = gym.make_vec("gym_copter_amd/Lander2D-v0", num_envs=2, autoreset_mode="disabled")
    assert env2.metadata["autoreset_mode"] is gym.vector.AutoresetMode.DISABLED and env2.action_space.shape == (2, 2)
    with pytest.raises(TypeError):
        gym.make_vec("gym_copter_amd/Lander-v0", num_envs=2, no_such_keyword=1)
    # the namespaced id as gym_copter_amd.make takes it ('gym_copter:Lander-v0' style, lander.py:80)
    env3 = gca.make("gym_copter_amd:Lander-v0", num_envs=5)
    assert env3.num_envs == 5 and env3.spec is None


def test_a_failing_registration_is_logged_not_swallowed(gca_with_gymnasium, caplog):
    gca, gym, _ = gca_with_gymnasium
    reg = gym.envs.registration

    def broken(**kw):
        raise ValueError("registry says no")
    reg.register = broken
    with caplog.at_level(logging.WARNING, logger="gym_copter_amd"):
        got = gca.register_with_gymnasium(namespace="other_ns")
    assert got == []
    msgs = [r.getMessage() for r in caplog.records if r.name == "gym_copter_amd"]
    assert len(msgs) == 1 and "7 of 7 ids could not be registered" in msgs[0] and "registry says no" in msgs[0]


def test_without_gymnasium_the_same_attributes_exist_on_a_plain_class():
    import gym_copter_amd as gca
    if gca.vecenv.HAVE_GYMNASIUM:
        pytest.skip("a real gymnasium is installed here")
    assert gca._GYMNASIUM_IDS == [] and gca.vecenv._VectorEnvBase is object
    real = gca.CopterVecEnv._open_device
    gca.CopterVecEnv._open_device = lambda self: None
    try:
        env = gca.make("Lander-v0", num_envs=4)
    finally:
        gca.CopterVecEnv._open_device = real
    mode = env.metadata["autoreset_mode"]
    assert (mode.name, mode.value) == ("NEXT_STEP", "NextStep")       # gymnasium.vector.AutoresetMode's spelling
    assert env.spec is None and env.render_mode is None and env.closed is False and env.unwrapped is env
    env.spec = "settable"
    assert isinstance(env.single_action_space, gca.Box) and env.action_space.shape == (4, 4)
    env.close()
    assert env.closed
