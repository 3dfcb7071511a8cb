"""TEST-ONLY stand-in for the `gymnasium` package.  NOT gymnasium.  NOT shipped.

Why it exists: `/root/reference/rware/warehouse.py:5-6` imports gymnasium at module
top, and gymnasium is not installed in the build container (no wheel, no network).
This directory is put on `sys.path` ONLY by `oracle/ref_runner.py` /
`tests/golden/generate_golden.py`, and ONLY when `import gymnasium` fails, so the
unmodified reference can run here and produce golden vectors.  It supplies exactly
the names the reference touches (`Env`, `spaces.*`, `utils.seeding.np_random`,
`register`, `Wrapper`).  The product package (`robotic-warehouse_amd/`) never
imports it.

`utils.seeding.np_random(seed)` follows Gymnasium's published definition
(`Generator(PCG64(SeedSequence(seed)))`); it could not be diffed against the real
package here.
"""
from . import spaces  # noqa: F401
from .utils import seeding

__version__ = "0.0-standin"
IS_STANDIN = True

registry = {}


def register(id, entry_point=None, kwargs=None, **_ignored):
    registry[id] = {"entry_point": entry_point, "kwargs": dict(kwargs or {})}


def make(id, **kwargs):
    import importlib

    spec = registry[id]
    mod_name, cls_name = spec["entry_point"].split(":")
    cls = getattr(importlib.import_module(mod_name), cls_name)
    kw = dict(spec["kwargs"])
    kw.update(kwargs)
    return cls(**kw)


class Env:
    metadata = {}
    _np_random = None
    _np_random_seed = None

    @property
    def np_random(self):
        if self._np_random is None:
            self._np_random, self._np_random_seed = seeding.np_random()
        return self._np_random

    @np_random.setter
    def np_random(self, value):
        self._np_random = value

    @property
    def unwrapped(self):
        return self

    def reset(self, *, seed=None, options=None):
        if seed is not None:
            self._np_random, self._np_random_seed = seeding.np_random(seed)

    def close(self):
        pass


class Wrapper(Env):
    def __init__(self, env):
        self.env = env
        self.action_space = getattr(env, "action_space", None)
        self.observation_space = getattr(env, "observation_space", None)

    @property
    def unwrapped(self):
        return self.env.unwrapped

    def reset(self, **kwargs):
        return self.env.reset(**kwargs)

    def step(self, action):
        return self.env.step(action)

    def __getattr__(self, name):
        return getattr(self.env, name)


class ObservationWrapper(Wrapper):
    def reset(self, **kwargs):
        obs, info = self.env.reset(**kwargs)
        return self.observation(obs), info

    def step(self, action):
        obs, r, d, t, info = self.env.step(action)
        return self.observation(obs), r, d, t, info
