"""Object-graph facade over a batched backend so the reference's state-injection tests
(/root/reference/tests/test_movement.py, test_goals.py, test_env.py: build a Warehouse, overwrite
`env.agents[i].x/.y/.dir/.carrying_shelf`, `env.shelfs[j].x/.y`, `env.request_queue[k]`, call
`_recalc_grid()`, step, assert) can be replayed against the oracle, the emulated engine and the
real HIP engine.  Every env of the batch gets the same injected state; results must agree across
the batch, and env 0 is returned."""
import numpy as np

import rware_amd
from rware_oracle import OracleVecEnv

UP, DOWN, LEFT, RIGHT = 0, 1, 2, 3
NOOP, FORWARD, TURN_LEFT, TURN_RIGHT, TOGGLE = 0, 1, 2, 3, 4
GLOBAL, INDIVIDUAL, TWO_STAGE = 0, 1, 2


class _Agent:
    def __init__(self):
        self.x = self.y = self.dir = 0
        self.carrying_shelf = None
        self.has_delivered = False


class _Shelf:
    def __init__(self, id_):
        self.id, self.x, self.y = id_, 0, 0


class KatEnv:
    """Positional args follow Warehouse(shelf_columns, column_height, shelf_rows, n_agents, msg_bits,
    sensor_range, request_queue_size, max_inactivity_steps, max_steps, reward_type)."""

    def __init__(self, backend, shelf_columns, column_height, shelf_rows, n_agents, msg_bits, sensor_range,
                 request_queue_size, max_inactivity_steps, max_steps, reward_type, batch=3, library=None,
                 static_geometry=None, **extra):
        # static_geometry=(E, T): replay the scenario on the EXACT-SHAPE kernel build of rware-small-4ag (the headline
        # task: 20 x 10 grid, 4 agents, queue 4) instead of the generic kernel.  The fixture's 29 x 10 grid is the same
        # warehouse with one more shelf row on top, so fixture row y maps to row y - 9 of the small grid for the two
        # bottom blocks (y >= 18) and to itself for the top block (y < 9); agents the scenario does not have are parked
        # on the left wall and sent NOOPs.
        self.static = static_geometry is not None
        self.n_real = n_agents
        geom = (4, 64)
        if self.static:
            assert backend != "oracle" and (shelf_columns, column_height, shelf_rows) == (3, 8, 3) and n_agents <= 4
            assert sensor_range == 1 and not msg_bits and not extra
            shelf_rows, n_agents, request_queue_size = 2, 4, 4
            geom = static_geometry
            batch = 16
        kw = dict(shelf_columns=shelf_columns, column_height=column_height, shelf_rows=shelf_rows,
                  n_agents=n_agents, msg_bits=msg_bits, sensor_range=sensor_range,
                  request_queue_size=request_queue_size, max_inactivity_steps=max_inactivity_steps,
                  max_steps=max_steps, reward_type=reward_type, **extra)
        self.B, self.N = batch, n_agents
        self.kind = backend
        if backend == "oracle":
            self.be = OracleVecEnv(batch, **kw)
        else:
            kw.pop("msg_bits")
            self.be = rware_amd.WarehouseVecEnv(batch, autoreset_mode="disabled", library=library,
                                                envs_per_workgroup=geom[0], threads_per_workgroup=geom[1], **kw)
            if self.static:
                assert self.be.engines[0].info.specialised == 1, "the exact-shape kernel build was not selected"
        self.grid_size = (self.be.H, self.be.W) if backend == "oracle" else self.be.grid_size
        self.goals = [(int(g[0]), self._y_out(int(g[1]))) for g in self.be.goals]
        self.n_agents = self.n_real

    # fixture row <-> row of the small grid (static mode only)
    def _y_in(self, y):
        if not self.static:
            return y
        assert y < 9 or y >= 18, "the middle block of the 29-row fixture has no counterpart in the 20-row grid"
        return y if y < 9 else y - 9

    def _y_out(self, y):
        return y if not self.static or y < 9 else y + 9

    _PARK = [(0, 5), (0, 6), (0, 7)]  # left wall, clear of every scenario

    # -- reference-style surface ------------------------------------------------------------
    def reset(self, seed=0):
        if self.kind == "oracle":
            self.be.seed(seed)
            for e in range(self.B):
                self.be.rng[e] = self.be.rng[0]   # identical envs across the batch
            self.be.reset()
        else:
            self.be.reset(seed=[seed] * self.B)
        self._pull(init=True)
        return self

    def _state(self):
        return self.be.get_state()

    def _pull(self, init=False):
        st = self._state()
        for k, v in st.items():
            assert all(np.array_equal(v[0], v[e]) for e in range(1, self.B)), f"batch disagrees on {k}"
        if init:
            S = int(st["grid"][0, 1].max())
            self.shelfs = [_Shelf(i + 1) for i in range(S)]
            self.agents = [_Agent() for _ in range(self.n_real)]
        g = st["grid"][0]
        ys, xs = np.nonzero(g[1])
        for y, x in zip(ys, xs):
            s = self.shelfs[g[1, y, x] - 1]
            s.x, s.y = int(x), self._y_out(int(y))
        for i, a in enumerate(self.agents):
            a.x, a.y, a.dir = int(st["agent_x"][0, i]), self._y_out(int(st["agent_y"][0, i])), int(st["agent_dir"][0, i])
            c = int(st["agent_carry"][0, i])
            a.carrying_shelf = self.shelfs[c - 1] if c else None
            a.has_delivered = bool(st["agent_delivered"][0, i])
        self.request_queue = [self.shelfs[i - 1] for i in st["queue"][0]]
        self.grid = g
        self._cur_steps = int(st["steps"][0])
        self._cur_inactive_steps = int(st["inactive"][0])

    def _recalc_grid(self):
        """Push the (possibly hand-edited) object graph into the backend, as Warehouse._recalc_grid does."""
        rep = lambda a: np.repeat(np.asarray(a, np.int32)[None], self.B, axis=0)
        pad = self.N - self.n_real  # parked extra agents of the static mode: facing the left wall, empty-handed
        fields = dict(
            agent_x=rep([a.x for a in self.agents] + [p[0] for p in self._PARK[:pad]]),
            agent_y=rep([self._y_in(a.y) for a in self.agents] + [p[1] for p in self._PARK[:pad]]),
            agent_dir=rep([a.dir for a in self.agents] + [LEFT] * pad),
            agent_carry=rep([a.carrying_shelf.id if a.carrying_shelf else 0 for a in self.agents] + [0] * pad),
            agent_delivered=rep([int(a.has_delivered) for a in self.agents] + [0] * pad),
            queue=rep([s.id for s in self.request_queue]),
        )
        sxy = rep([[s.x, self._y_in(s.y)] for s in self.shelfs])
        if self.kind == "oracle":
            self.be.set_state(**fields)
            self.be.recalc_grid(sxy)
        else:
            self.be.set_state(refresh_obs=False, **fields)
            self.be.recalc_grid(sxy)

    def step(self, actions):
        self._recalc_grid()
        assert len(actions) == self.n_real
        acts = [int(getattr(x, "value", x)) for x in actions] + [NOOP] * (self.N - self.n_real)
        a = np.repeat(np.asarray(acts, np.int32)[None], self.B, axis=0)
        if self.kind == "oracle":
            rew, done = self.be.step(a)
            obs = self.be.obs()
        else:
            obs, rew, done, trunc, _ = self.be.step(a)
            assert not trunc.any()
        assert all(np.array_equal(obs[0], obs[e]) for e in range(1, self.B))
        self._pull()
        return obs[0][:self.n_real], [float(r) for r in rew[0][:self.n_real]], bool(done[0]), False, {}

    def close(self):
        if self.kind != "oracle":
            self.be.close()
