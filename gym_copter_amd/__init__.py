"""gym_copter_amd -- MI355X-native batch stepper for the gym-copter rigid-body hot path.

Only what the hot path needs lives here: csrc/ (HIP kernels + C ABI), the ctypes binding
and the Gymnasium-style vector-env host class.  See DESIGN.md and INTEGRATION.md.
"""
from ._lib import CopterStepError  # noqa: F401
from .spaces import Box  # noqa: F401
from .vecenv import CopterVecEnv  # noqa: F401

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
