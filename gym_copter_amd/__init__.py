"""gym_copter_amd -- MI355X-native batch stepper for the gym-copter rigid-body hot path.

Only what the hot path needs lives here: csrc/ (HIP kernels + C ABI), the ctypes binding
and the Gymnasium-style vector-env host class.  See DESIGN.md and INTEGRATION.md.
"""
from ._lib import CopterStepError  # noqa: F401
from .vecenv import CopterVecEnv  # noqa: F401
from .spaces import box_class as _box_class

# The Box class the envs' spaces are instances of: gymnasium.spaces.Box when Gymnasium is importable (so that
# isinstance(env.single_action_space, gym_copter_amd.Box) and Box.__eq__ hold there too), else the minimal one of spaces.py
Box = _box_class()
from .policy_jit import compile_policy, load_policy  # noqa: F401

__version__ = "0.1.0"

# Environment ids: 'Lander-v0' is the live upstream registration
# (reference gym_copter/__init__.py:9-13, max_episode_steps=1000); 'Lander3D-v0' and
# 'Hover3D-v0' are the names the retired upstream tree used (attic/gym_copter/__init__.py).
_REGISTRY = {
    "Lander-v0": dict(task="lander3d", max_steps=1000),
    "Lander3D-v0": dict(task="lander3d", max_steps=1000),
    "Hover3D-v0": dict(task="hover3d", max_steps=1000),
    "Lander2D-v0": dict(task="lander2d", max_steps=1000),
    "Lander1D-v0": dict(task="lander1d", max_steps=1000),
    "Hover2D-v0": dict(task="hover2d", max_steps=1000),
    "Hover1D-v0": dict(task="hover1d", max_steps=1000),
}


def make(env_id, num_envs=1, **kwargs):
    """Batched counterpart of gym.make('gym_copter:Lander-v0')."""
    env_id = env_id.split(":")[-1]
    if env_id not in _REGISTRY:
        raise KeyError("unknown environment id %r (have %s)" % (env_id, sorted(_REGISTRY)))
    spec = dict(_REGISTRY[env_id])
    spec.update(kwargs)
    return CopterVecEnv(num_envs=num_envs, **spec)


def make_vec(env_id, num_envs=1, **kwargs):
    return make(env_id, num_envs=num_envs, **kwargs)


def register_with_gymnasium(namespace="gym_copter_amd"):
    """Register the ids above with Gymnasium when it is importable, as the reference registers 'Lander-v0'
    (reference gym_copter/__init__.py:9-13: `register(id='Lander-v0', entry_point=..., max_episode_steps=1000)`, reached
    through gym.make('gym_copter:Lander-v0'), lander.py:80): gymnasium.make_vec('gym_copter_amd/Lander-v0', num_envs=N)
    then builds a CopterVecEnv through `vector_entry_point` (make_vec calls it as
    entry(num_envs=N, max_episode_steps=1000, **spec kwargs, **make_vec kwargs)).  Returns the ids registered.

    Gymnasium is not a dependency: without it the result is [] and nothing is logged.  With it, a registration that
    fails is logged once (logger 'gym_copter_amd') instead of passing silently; `gym_copter_amd.make` works either way."""
    import logging
    log = logging.getLogger("gym_copter_amd")
    try:
        from gymnasium.envs.registration import register, registry
    except ImportError:
        return []
    except Exception as e:       # a Gymnasium that is there but broken: say so
        log.warning("gym_copter_amd: gymnasium is installed but its registry could not be imported (%r); "
                    "gymnasium.make_vec ids are not registered, gym_copter_amd.make still works", e)
        return []
    done, failed = [], []
    for env_id, spec in _REGISTRY.items():
        full = "%s/%s" % (namespace, env_id)
        if full in registry:
            done.append(full)
            continue
        try:
            register(id=full, vector_entry_point="gym_copter_amd:_vector_entry_point",
                     max_episode_steps=spec["max_steps"], kwargs={"copter_id": env_id})
            done.append(full)
        except Exception as e:
            failed.append((full, e))
    if failed:
        log.warning("gym_copter_amd: %d of %d ids could not be registered with gymnasium (first: %s: %r); "
                    "gym_copter_amd.make still works", len(failed), len(_REGISTRY), failed[0][0], failed[0][1])
    return done


def _vector_entry_point(copter_id=None, num_envs=1, max_episode_steps=None, **kwargs):
    """What gymnasium.make_vec calls for the ids registered above.  `max_episode_steps` (the registration's 1000, or
    the caller's make_vec(..., max_episode_steps=K)) is the env's own step limit `max_steps` (task.py:36, :128-129):
    the batch env applies the limit itself, as Gymnasium expects of a vector entry point (no TimeLimit wrapper is
    put around it)."""
    if copter_id is None:
        raise TypeError("_vector_entry_point needs copter_id (it is part of the registered kwargs)")
    if max_episode_steps is not None and "max_steps" not in kwargs:
        kwargs["max_steps"] = int(max_episode_steps)
    return make(copter_id, num_envs=num_envs, **kwargs)


_GYMNASIUM_IDS = register_with_gymnasium()
