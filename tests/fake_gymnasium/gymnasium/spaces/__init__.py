"""TEST INFRASTRUCTURE — fake of gymnasium.spaces: shapes, dtypes and `contains` (see ../__init__.py)."""
import numpy as np


class Space:
    def __init__(self, shape=None, dtype=None):
        self.shape = None if shape is None else tuple(shape)
        self.dtype = None if dtype is None else np.dtype(dtype)

    def __contains__(self, x):
        return self.contains(x)


class Box(Space):
    def __init__(self, low, high, shape=None, dtype=np.float32):
        super().__init__(np.shape(low) if shape is None else shape, dtype)
        self.low = np.full(self.shape, low, dtype=self.dtype)
        self.high = np.full(self.shape, high, dtype=self.dtype)

    def contains(self, x):
        if not isinstance(x, np.ndarray):
            try:
                x = np.asarray(x, dtype=self.dtype)
            except (ValueError, TypeError):
                return False
        return bool(np.can_cast(x.dtype, self.dtype) and x.shape == self.shape and np.all(x >= self.low) and np.all(x <= self.high))


class Discrete(Space):
    def __init__(self, n, start=0):
        super().__init__((), np.int64)
        self.n, self.start = int(n), int(start)

    def contains(self, x):
        if isinstance(x, (np.generic, np.ndarray)):
            if not (np.issubdtype(np.asarray(x).dtype, np.integer) and np.asarray(x).shape == ()):
                return False
            x = int(x)
        elif not isinstance(x, int):
            return False
        return self.start <= x < self.start + self.n


class MultiDiscrete(Space):
    def __init__(self, nvec, dtype=np.int64):
        self.nvec = np.array(nvec, dtype=dtype, copy=True)
        super().__init__(self.nvec.shape, dtype)

    def contains(self, x):
        x = np.asarray(x)
        return bool(x.shape == self.shape and np.issubdtype(x.dtype, np.integer) and np.all(x >= 0) and np.all(x < self.nvec))


class Tuple(Space):
    def __init__(self, spaces):
        super().__init__(None, None)
        self.spaces = tuple(spaces)

    def __len__(self):
        return len(self.spaces)

    def __getitem__(self, i):
        return self.spaces[i]

    def contains(self, x):
        if isinstance(x, (list, np.ndarray)):
            x = tuple(x)
        return isinstance(x, tuple) and len(x) == len(self.spaces) and all(s.contains(p) for s, p in zip(self.spaces, x))
