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
