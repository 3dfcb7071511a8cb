"""TEST INFRASTRUCTURE — drives the UNMODIFIED reference (`/root/reference/rware`) in the
build container to (a) pin the CPU oracle and (b) generate `tests/golden/*.npz`.

On the GPU box `/root/reference` does not exist; the only thing that runs there is `bench.py`'s
`cpu_baseline` leg timing the staged, unmodified copy under the git-ignored `oracle/_ref/`
(`oracle/make_ref.sh`).  The product package never imports this module.

Pinned tie-break.  The reference resolves a tree-shaped collision component with
`nx.algorithms.dag_longest_path(comp)` (`rware/warehouse.py:865`).  When two
predecessors of a cell have equal chain depth, networkx keeps the *first* maximal one
in `G.pred[v]` order, and that order comes from CPython set iteration inside
`G.subgraph(c).copy()` — not a portable rule.  BASELINE.json words
parity as "identical seeds and tie-break rule", so the rule is pinned on both sides:

    among equal-depth predecessors of a cell, the LOWEST AGENT ID wins.

`pinned_tiebreak()` monkey-patches exactly that one function with a variant that is
identical except for the tie rule.  `TieStats` additionally records how often the
unpatched networkx choice differs (reported in DESIGN.md as a statistic).
"""
from __future__ import annotations

import contextlib
import os
import sys
from collections import deque

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_STANDIN = os.path.join(_HERE, "gymnasium_standin")
STAGED_ROOT = os.path.join(_HERE, "_ref")  # oracle/make_ref.sh: byte-for-byte copy of rware/{__init__,warehouse}.py (git-ignored)


def _find_reference_root() -> str:
    """/root/reference (build container) first; else the staged copy under oracle/_ref (what the GPU box gets)."""
    cands = [os.environ.get("RWARE_REFERENCE_ROOT"), "/root/reference", STAGED_ROOT]
    for c in cands:
        if c and os.path.isfile(os.path.join(c, "rware", "warehouse.py")):
            return c
    return cands[0] or "/root/reference"


REFERENCE_ROOT = _find_reference_root()

_rware = None
_CURRENT_ENV = None


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "rware", "warehouse.py"))


def load_reference():
    """Import `rware.warehouse` from /root/reference (stand-in gymnasium if the real one is absent)."""
    global _rware
    if _rware is not None:
        return _rware
    if not reference_available():
        raise RuntimeError(f"reference not found under {REFERENCE_ROOT}")
    try:
        import gymnasium  # noqa: F401
    except ImportError:
        sys.path.insert(0, _STANDIN)
        import gymnasium  # noqa: F401
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import rware.warehouse as wh  # noqa: E402

    _rware = wh
    return wh


def using_standin_gymnasium() -> bool:
    import gymnasium

    return bool(getattr(gymnasium, "IS_STANDIN", False))


# --------------------------------------------------------------------------------------
# pinned tie-break
# --------------------------------------------------------------------------------------
class TieStats:
    calls = 0          # dag_longest_path invocations
    ties = 0           # invocations in which some cell had >=2 equal-depth best predecessors
    disagree = 0       # invocations where unpatched networkx committed a different agent set

    @classmethod
    def reset(cls):
        cls.calls = cls.ties = cls.disagree = 0


def _dag_longest_path_pinned(G, weight="weight", default_weight=1, topo_order=None):
    """networkx.algorithms.dag.dag_longest_path with the pinned tie rule (lowest agent id)."""
    import networkx as nx

    if not G:
        return []
    env = _CURRENT_ENV
    grid_agents = env.grid[0]

    def aid(node):
        return int(grid_agents[node[1], node[0]])

    dist = {}
    tie = False
    for v in nx.topological_sort(G):
        preds = list(G.pred[v])
        if preds:
            best_d = max(dist[u][0] for u in preds)
            cands = [u for u in preds if dist[u][0] == best_d]
            if len(cands) > 1:
                tie = True
            u = min(cands, key=aid)
            dist[v] = (best_d + 1, u)
        else:
            dist[v] = (0, v)
    u = None
    v = max(dist, key=lambda x: dist[x][0])
    path = []
    while u != v:
        path.append(v)
        u = v
        v = dist[v][1]
    path.reverse()
    TieStats.calls += 1
    if tie:
        TieStats.ties += 1
        orig = _ORIG_DLP(G)
        if {aid(n) for n in orig} != {aid(n) for n in path}:
            TieStats.disagree += 1
    return path


_ORIG_DLP = None


@contextlib.contextmanager
def pinned_tiebreak():
    import networkx as nx

    global _ORIG_DLP
    _ORIG_DLP = nx.algorithms.dag_longest_path
    nx.algorithms.dag_longest_path = _dag_longest_path_pinned
    try:
        yield
    finally:
        nx.algorithms.dag_longest_path = _ORIG_DLP


def ref_step(env, actions):
    """`env.step` under the pinned tie-break."""
    global _CURRENT_ENV
    _CURRENT_ENV = env
    with pinned_tiebreak():
        return env.step(list(actions))


class _CountingGenerator:
    """Stands in front of the env's numpy Generator for ONE step and counts its `choice` calls: `Warehouse.step` draws exactly one
    replacement request per delivered shelf (`rware/warehouse.py:916`) and nothing else.  Every other attribute is the Generator's."""

    def __init__(self, gen):
        self._gen, self.choice_calls = gen, 0

    def choice(self, *args, **kwargs):
        self.choice_calls += 1
        return self._gen.choice(*args, **kwargs)

    def __getattr__(self, name):
        return getattr(self._gen, name)


def ref_step_events(env, actions):
    """`ref_step` plus the two event counts the engine can keep per env (RW_BUF_STAT_*), read off the unmodified reference as it runs:
    deliveries = replacement draws of this step (`rware/warehouse.py:907-917`: one `np_random.choice` per delivered shelf — counting
    changed queue slots instead would miss a slot that is refilled twice in one step, when the second goal holds the shelf the first
    delivery has just requested), failed moves = agents that asked for FORWARD and whose `req_action` the step turned into NOOP
    (`:843-846` shelf-block cancel, `:871-876` collision resolution).  Returns (step result, deliveries, failed moves)."""
    wh = load_reference()
    asked = [wh.Action(a[0] if env.msg_bits > 0 else a) for a in actions]
    gen = env.np_random
    counting = _CountingGenerator(gen)
    env.np_random = counting
    try:
        out = ref_step(env, actions)
    finally:
        env.np_random = gen
    failed = sum(1 for ag, a in zip(env.agents, asked) if a == wh.Action.FORWARD and ag.req_action == wh.Action.NOOP)
    return out, counting.choice_calls, failed


# --------------------------------------------------------------------------------------
# registry kwargs (read from the reference's own register() calls) and construction
# --------------------------------------------------------------------------------------
def registry_kwargs(env_id: str) -> dict:
    load_reference()
    import gymnasium
    import rware  # noqa: F401  (runs the register() calls, rware/__init__.py:22-39)

    reg = getattr(gymnasium, "registry")
    spec = reg[env_id]
    kw = spec["kwargs"] if isinstance(spec, dict) else dict(spec.kwargs)
    return dict(kw)


def make_reference_env(env_id: str | None = None, **kwargs):
    wh = load_reference()
    kw = registry_kwargs(env_id) if env_id else {}
    kw.update(kwargs)
    return wh.Warehouse(**kw)


# --------------------------------------------------------------------------------------
# state snapshot in the SoA form the oracle / engine use
# --------------------------------------------------------------------------------------
def rng_state_tuple(env):
    st = env.np_random.bit_generator.state
    s, inc = st["state"]["state"], st["state"]["inc"]
    m = (1 << 64) - 1
    return np.array(
        [s >> 64, s & m, inc >> 64, inc & m, st["has_uint32"], st["uinteger"]], dtype=np.uint64
    )


def snapshot(env) -> dict:
    n = env.n_agents
    out = {
        "grid": env.grid.astype(np.int32).copy(),
        "agent_x": np.array([a.x for a in env.agents], dtype=np.int32),
        "agent_y": np.array([a.y for a in env.agents], dtype=np.int32),
        "agent_dir": np.array([a.dir.value for a in env.agents], dtype=np.int32),
        "agent_carry": np.array(
            [a.carrying_shelf.id if a.carrying_shelf else 0 for a in env.agents], dtype=np.int32
        ),
        "agent_delivered": np.array([int(a.has_delivered) for a in env.agents], dtype=np.int32),
        "queue": np.array([s.id for s in env.request_queue], dtype=np.int32),
        "steps": np.int32(env._cur_steps),
        "inactive": np.int32(env._cur_inactive_steps),
        "rng": rng_state_tuple(env),
    }
    if env.msg_bits:
        out["agent_msg"] = np.array([sum(int(b) << k for k, b in enumerate(a.message)) for a in env.agents], dtype=np.int32)
    assert out["agent_x"].shape == (n,)
    return out


def obs_array(obs_tuple) -> np.ndarray:
    return np.stack([np.asarray(o, dtype=np.float32) for o in obs_tuple])


# --------------------------------------------------------------------------------------
# a delivery-seeking scripted policy (random play almost never delivers, so the
# request-replacement RNG draw of rware/warehouse.py:915-917 would go untested)
# --------------------------------------------------------------------------------------
_DIRV = {0: (0, -1), 1: (0, 1), 2: (-1, 0), 3: (1, 0)}  # Direction value -> (dx, dy)


def _bfs_next(env, agent, targets, loaded):
    """First move (dx,dy) of a shortest path from agent to any target cell; None if unreachable."""
    H, W = env.grid_size
    start = (agent.x, agent.y)
    if start in targets:
        return (0, 0)
    prev = {start: None}
    dq = deque([start])
    while dq:
        c = dq.popleft()
        for dx, dy in _DIRV.values():
            nx_, ny_ = c[0] + dx, c[1] + dy
            if not (0 <= nx_ < W and 0 <= ny_ < H) or (nx_, ny_) in prev:
                continue
            if loaded and env.grid[1, ny_, nx_] != 0:
                continue
            prev[(nx_, ny_)] = c
            if (nx_, ny_) in targets:
                node = (nx_, ny_)
                while prev[node] != start:
                    node = prev[node]
                return (node[0] - start[0], node[1] - start[1])
            dq.append((nx_, ny_))
    return None


def scripted_actions(env, rng: np.random.Generator, eps: float = 0.15):
    wh = load_reference()
    A = wh.Action
    requested = {s.id: s for s in env.request_queue}
    carried = {a.carrying_shelf.id for a in env.agents if a.carrying_shelf}
    acts = []
    for ag in env.agents:
        if rng.random() < eps:
            acts.append(int(rng.integers(0, 5)))
            continue
        if ag.carrying_shelf is not None:
            if ag.carrying_shelf.id in requested:
                targets, loaded = set(env.goals), True
            else:
                if not env.highways[ag.y, ag.x]:
                    acts.append(A.TOGGLE_LOAD.value)
                    continue
                H, W = env.grid_size
                targets = {
                    (x, y)
                    for y in range(H)
                    for x in range(W)
                    if not env.highways[y, x] and env.grid[1, y, x] == 0
                }
                loaded = True
        else:
            targets = {(s.x, s.y) for sid, s in requested.items() if sid not in carried}
            loaded = False
            if (ag.x, ag.y) in targets:
                acts.append(A.TOGGLE_LOAD.value)
                continue
        mv = _bfs_next(env, ag, targets, loaded) if targets else None
        if mv is None or mv == (0, 0):
            acts.append(int(rng.integers(0, 5)))
            continue
        want = [d for d, v in _DIRV.items() if v == mv][0]
        if ag.dir.value == want:
            acts.append(A.FORWARD.value)
        else:
            wrap = [0, 3, 1, 2]  # UP, RIGHT, DOWN, LEFT  (rware/warehouse.py:119)
            ci, wi = wrap.index(ag.dir.value), wrap.index(want)
            acts.append(A.RIGHT.value if (wi - ci) % 4 in (1, 2) else A.LEFT.value)
    return acts
