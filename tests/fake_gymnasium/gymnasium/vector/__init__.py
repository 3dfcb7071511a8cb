"""TEST INFRASTRUCTURE — fake of gymnasium.vector (see ../__init__.py)."""
import enum

from ..utils import seeding


class AutoresetMode(enum.Enum):
    NEXT_STEP = "NextStep"
    SAME_STEP = "SameStep"
    DISABLED = "Disabled"


class VectorEnv:
    metadata = {}
    spec = None
    render_mode = None
    closed = False
    observation_space = None
    action_space = None
    single_observation_space = None
    single_action_space = None
    num_envs = None
    _np_random = None
    _np_random_seed = None

    def reset(self, *, seed=None, options=None):
        if seed is not None:
            self._np_random, self._np_random_seed = seeding.np_random(seed)

    def step(self, actions):
        raise NotImplementedError

    def render(self):
        raise NotImplementedError

    def close(self, **kwargs):
        if self.closed:
            return
        self.close_extras(**kwargs)
        self.closed = True

    def close_extras(self, **kwargs):
        pass

    @property
    def np_random(self):
        if self._np_random is None:
            self._np_random, self._np_random_seed = seeding.np_random()
        return self._np_random

    @np_random.setter
    def np_random(self, value):
        self._np_random = value
        self._np_random_seed = -1

    @property
    def unwrapped(self):
        return self

    def __del__(self):
        if not getattr(self, "closed", True):
            self.close()
