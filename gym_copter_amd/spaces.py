"""Minimal Box space with the attributes callers of the reference rely on
(`shape`, `dtype`, `low`, `high`, `sample`, `contains`); mirrors
gymnasium.spaces.Box as used at reference envs/task.py:46-55.  gymnasium itself is
not a dependency."""
import numpy as np


class Box:
    def __init__(self, low, high, shape, dtype=np.float32, seed=None):
        self.shape = tuple(shape)
        self.dtype = np.dtype(dtype)
        self.low = np.full(self.shape, low, dtype=self.dtype)
        self.high = np.full(self.shape, high, dtype=self.dtype)
        self._rng = np.random.default_rng(seed)

    def seed(self, seed=None):
        self._rng = np.random.default_rng(seed)
        return [seed]

    def sample(self):
        lo = np.where(np.isfinite(self.low), self.low, -1e6)
        hi = np.where(np.isfinite(self.high), self.high, 1e6)
        return self._rng.uniform(lo, hi).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return bool(x.shape == self.shape and np.all(x >= self.low) and np.all(x <= self.high))

    __contains__ = contains

    def __repr__(self):
        return "Box(%s, %s, %s, %s)" % (self.low.min(), self.high.max(), self.shape, self.dtype.name)

    def __eq__(self, other):
        return (isinstance(other, Box) and self.shape == other.shape and self.dtype == other.dtype
                and np.array_equal(self.low, other.low) and np.array_equal(self.high, other.high))


def batch_space(space, n):
    return Box(space.low.flat[0], space.high.flat[0], (n,) + space.shape, space.dtype)


def box_class():
    """The class of the spaces gymnasium_api()'s `box` builds: gymnasium.spaces.Box if importable, else Box above."""
    try:
        from gymnasium.spaces import Box as GBox
        return GBox
    except ImportError:
        return Box


class _OwnAutoresetMode:
    """Stand-in for gymnasium.vector.AutoresetMode (same member names and values) for hosts without gymnasium."""

    def __init__(self, name, value):
        self.name, self.value = name, value

    def __repr__(self):
        return "<AutoresetMode.%s: %r>" % (self.name, self.value)


def gymnasium_api():
    """-> (VectorEnv base class, Box, batch_space, {mode name: AutoresetMode member}, have_gymnasium).

    With Gymnasium importable these are ITS classes -- gymnasium.vector.VectorEnv as the base of CopterVecEnv,
    gymnasium.spaces.Box spaces, gymnasium.vector.utils.batch_space and gymnasium.vector.AutoresetMode members for
    metadata["autoreset_mode"] (what gymnasium.make_vec checks, Gymnasium >= 1.1) -- so that isinstance checks of
    VectorEnv consumers hold.  Without it (gymnasium is not a dependency): `object`, the minimal Box above and
    look-alike mode objects with the same names and values."""
    try:
        from gymnasium.vector import VectorEnv
        from gymnasium.spaces import Box as GBox
        try:
            from gymnasium.vector.utils import batch_space as gbatch
        except Exception:      # older layouts
            gbatch = None
        try:
            from gymnasium.vector import AutoresetMode as GMode
            modes = {"next_step": GMode.NEXT_STEP, "same_step": GMode.SAME_STEP, "disabled": GMode.DISABLED}
        except Exception:      # Gymnasium < 1.1: no such enum; plain look-alikes
            modes = None

        def box(low, high, shape, dtype=np.float32):
            return GBox(low, high, shape=tuple(shape), dtype=dtype)

        def batch(space, n):
            if gbatch is not None:
                return gbatch(space, n)
            return GBox(space.low.flat[0], space.high.flat[0], shape=(n,) + tuple(space.shape), dtype=space.dtype)
        have = True
    except ImportError:
        VectorEnv, box, batch, modes, have = object, Box, batch_space, None, False
    if modes is None:
        modes = {"next_step": _OwnAutoresetMode("NEXT_STEP", "NextStep"),
                 "same_step": _OwnAutoresetMode("SAME_STEP", "SameStep"),
                 "disabled": _OwnAutoresetMode("DISABLED", "Disabled")}
    return VectorEnv, box, batch, modes, have
