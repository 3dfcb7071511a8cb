"""TEST-ONLY stand-in (see ../__init__.py) for gymnasium.spaces.

Only what `rware/warehouse.py` constructs and `flatdim`s; `sample()` exists so the
golden generator can draw actions.  No `contains`/`flatten` fidelity is claimed.
"""
from collections import OrderedDict

import numpy as np


class Space:
    def __init__(self, shape=None, dtype=None):
        self.shape = shape
        self.dtype = dtype
        self._rng = np.random.default_rng(0)

    def seed(self, seed=None):
        self._rng = np.random.default_rng(seed)


class Discrete(Space):
    def __init__(self, n, start=0):
        super().__init__((), np.int64)
        self.n = int(n)
        self.start = int(start)

    def sample(self):
        return int(self.start + self._rng.integers(self.n))


class MultiDiscrete(Space):
    def __init__(self, nvec):
        self.nvec = np.asarray(nvec, dtype=np.int64)
        super().__init__(self.nvec.shape, np.int64)

    def sample(self):
        return (self._rng.random(self.nvec.shape) * self.nvec).astype(np.int64)


class MultiBinary(Space):
    def __init__(self, n):
        self.n = n
        super().__init__((n,) if np.isscalar(n) else tuple(n), np.int8)

    def sample(self):
        return self._rng.integers(0, 2, size=self.shape, dtype=np.int8)


class Box(Space):
    def __init__(self, low, high, shape=None, dtype=np.float32):
        if shape is None:
            shape = np.asarray(low).shape
        self.low = np.broadcast_to(np.asarray(low, dtype=np.float64), shape)
        self.high = np.broadcast_to(np.asarray(high, dtype=np.float64), shape)
        super().__init__(tuple(shape), dtype)


class Tuple(Space):
    def __init__(self, spaces):
        self.spaces = tuple(spaces)
        super().__init__(None, None)

    def __iter__(self):
        return iter(self.spaces)

    def __len__(self):
        return len(self.spaces)

    def __getitem__(self, i):
        return self.spaces[i]

    def sample(self):
        return tuple(s.sample() for s in self.spaces)

    def seed(self, seed=None):
        for i, s in enumerate(self.spaces):
            s.seed(None if seed is None else seed + i)


class Dict(Space):
    def __init__(self, spaces=None, **kw):
        if spaces is None:
            spaces = kw
        self.spaces = OrderedDict(spaces)
        super().__init__(None, None)

    def __getitem__(self, k):
        return self.spaces[k]

    def keys(self):
        return self.spaces.keys()


def flatdim(space):
    if isinstance(space, Box):
        return int(np.prod(space.shape))
    if isinstance(space, Discrete):
        return space.n
    if isinstance(space, MultiBinary):
        return int(np.prod(space.shape))
    if isinstance(space, MultiDiscrete):
        return int(np.sum(space.nvec))
    if isinstance(space, Tuple):
        return sum(flatdim(s) for s in space.spaces)
    if isinstance(space, Dict):
        return sum(flatdim(s) for s in space.spaces.values())
    raise NotImplementedError(type(space))
