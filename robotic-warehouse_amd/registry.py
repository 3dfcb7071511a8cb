"""The reference's registry ids, preserved.

`import rware` registers 228 ids `rware-{tiny,small,medium,large}-{1..19}ag{-easy,,-hard}-v2`
(rware/__init__.py:22-39); README.md:88 and BASELINE.json spell the same tasks `-v1`.  Both
suffixes resolve here to the same constructor kwargs.
"""
from __future__ import annotations

import re

from .enums import RewardType

_SIZES = {"tiny": (1, 3), "small": (2, 3), "medium": (2, 5), "large": (3, 5)}  # (shelf_rows, shelf_columns)
_DIFFICULTY = {"-easy": 2, "": 1, "-hard": 0.5}
_ID = re.compile(r"^rware-(tiny|small|medium|large)-(\d+)ag(-easy|-hard)?-v([12])$")
# the opt-in registries of the reference: image_registration() (rware/__init__.py:42-80) and full_registration() (:83-175)
_OBS = r"(-img|-imgdict)?(-Nd)?"
_ID_IMAGE = re.compile(rf"^rware{_OBS}-(tiny|small|medium|large)-(\d+)ag(-easy|-hard)?-v([12])$")
_ID_FULL_SIZE = re.compile(rf"^rware{_OBS}(-[2-5]s)?-(tiny|small|medium|large)-(\d+)h-(\d+)ag(-easy|-hard)?-v([12])$")
_ID_FULL_GRID = re.compile(rf"^rware{_OBS}(-[2-5]s)?-(\d+)x(\d+)-(\d+)h-(\d+)ag-(\d+)req-(indiv|global|twostage)-v([12])$")
_REWARDS = {"indiv": RewardType.INDIVIDUAL, "global": RewardType.GLOBAL, "twostage": RewardType.TWO_STAGE}


def _base(rows, cols, height, agents, queue, sensor_range=1, reward=RewardType.INDIVIDUAL):
    return {
        "column_height": height, "shelf_rows": rows, "shelf_columns": cols, "n_agents": agents, "msg_bits": 0,
        "sensor_range": sensor_range, "request_queue_size": queue, "max_inactivity_steps": None, "max_steps": 500,
        "reward_type": reward,
    }


def _obs_kwargs(obs, nd, allow_imgdict, allow_bare_nd=False):
    """`-img` / `-imgdict` / `-Nd` id parts -> observation kwargs, with the reference's own restrictions."""
    from .enums import ObservationType

    if nd and not obs:
        if allow_bare_nd:  # the rows x cols loop of full_registration() has no such filter (:143-175): the id exists, FLATTENED
            return {"image_observation_directional": False}
        raise KeyError("-Nd ids exist only with image observations")  # (:59-61, :108-110)
    if obs == "-imgdict" and not allow_imgdict:
        raise KeyError("full_registration() has no -imgdict ids")     # (:84)
    if not obs:
        return {}
    return {"observation_type": ObservationType.IMAGE if obs == "-img" else ObservationType.IMAGE_DICT,
            "image_observation_directional": not nd}


def env_kwargs(env_id: str) -> dict:
    """Constructor kwargs of a reference id: the 228 ids registered at import (rware/__init__.py:22-39) and every id
    `image_registration()` / `full_registration()` would add — same grammar, same kwargs, `-v1` and `-v2`."""
    m = _ID.match(env_id)
    if m:
        if not 1 <= int(m.group(2)) <= 19:
            raise KeyError(f"unknown rware id {env_id!r}")
        size, agents, diff = m.group(1), int(m.group(2)), m.group(3) or ""
        return _base(_SIZES[size][0], _SIZES[size][1], 8, agents, int(agents * _DIFFICULTY[diff]))
    try:
        m = _ID_IMAGE.match(env_id)
        if m:  # rware{-img|-imgdict}{-Nd}-{size}-{agents}ag{diff}
            obs, nd, size, agents, diff = m.group(1) or "", m.group(2) or "", m.group(3), int(m.group(4)), m.group(5) or ""
            if not 1 <= agents <= 19:
                raise KeyError("agents")
            return dict(_base(_SIZES[size][0], _SIZES[size][1], 8, agents, int(agents * _DIFFICULTY[diff])),
                        **_obs_kwargs(obs, nd, True))
        m = _ID_FULL_SIZE.match(env_id)
        if m:  # rware{-img}{-Nd}{-Ns}-{size}-{h}h-{agents}ag{diff}
            obs, nd, sr, size = m.group(1) or "", m.group(2) or "", m.group(3), m.group(4)
            height, agents, diff = int(m.group(5)), int(m.group(6)), m.group(7) or ""
            if not (1 <= agents <= 19 and 1 <= height <= 15):
                raise KeyError("range")
            return dict(_base(_SIZES[size][0], _SIZES[size][1], height, agents, int(agents * _DIFFICULTY[diff]),
                              int(sr[1]) if sr else 1), **_obs_kwargs(obs, nd, False))
        m = _ID_FULL_GRID.match(env_id)
        if m:  # rware{-img}{-Nd}{-Ns}-{rows}x{cols}-{h}h-{agents}ag-{req}req-{reward}
            obs, nd, sr = m.group(1) or "", m.group(2) or "", m.group(3)
            rows, cols, height, agents, req = (int(m.group(k)) for k in (4, 5, 6, 7, 8))
            if not (1 <= rows <= 4 and cols in (3, 5, 7, 9) and 1 <= height <= 15 and 1 <= agents <= 19 and 1 <= req <= 19):
                raise KeyError("range")
            return dict(_base(rows, cols, height, agents, req, int(sr[1]) if sr else 1, _REWARDS[m.group(9)]),
                        **_obs_kwargs(obs, nd, False, allow_bare_nd=True))
    except KeyError:
        pass
    raise KeyError(f"unknown rware id {env_id!r}")


def all_ids(versions=("v1", "v2")):
    return [f"rware-{s}-{a}ag{d}-{v}" for v in versions for s in _SIZES for d in _DIFFICULTY for a in range(1, 20)]


def shard_seeds(rank: int, envs_per_rank: int, seed: int = 0):
    """Seeds of rank `rank`'s contiguous env shard when a batch is split over one process per GPU: env i of the
    rank is GLOBAL env rank*envs_per_rank + i and gets `seed + global index` — the Gymnasium vector convention
    (`reset(seed=s)` seeds env i with s + i) applied to the whole job, so the trajectories do not depend on how
    many GPUs the batch is spread over.  Pass to `Engine.reset(seeds=...)` / `WarehouseVecEnv.reset(seed=...)`."""
    import numpy as np

    lo = np.uint64(int(seed)) + np.uint64(int(rank) * int(envs_per_rank))
    return lo + np.arange(int(envs_per_rank), dtype=np.uint64)


def make_vec(env_id: str, num_envs: int, **kwargs):
    """`gym.make_vec(id, num_envs=B)` analogue that works without gymnasium installed."""
    from .vector_env import WarehouseVecEnv

    kw = env_kwargs(env_id)
    kw.update(kwargs)
    return WarehouseVecEnv(num_envs, **kw)


class Pipeline:
    """One sub-batch of `make_pipelines`: `env` (a WarehouseVecEnv over envs [lo, hi) of the whole batch, output="torch") and the
    torch `stream` everything of this sub-batch runs on.  `with pipe:` makes that stream torch's current one."""

    def __init__(self, env, stream, lo, hi):
        self.env, self.stream, self.lo, self.hi = env, stream, lo, hi
        self._ctx = None

    def __enter__(self):
        import torch

        self._ctx = torch.cuda.stream(self.stream)
        self._ctx.__enter__()
        return self.env

    def __exit__(self, *exc):
        ctx, self._ctx = self._ctx, None
        return ctx.__exit__(*exc)

    def reset(self, seed=None, **kw):
        """reset of this sub-batch with the seeds its envs have in the whole batch (env i: seed + i)."""
        with self:
            return self.env.reset(seed=None if seed is None else int(seed) + self.lo, **kw)


def make_pipelines(num_envs: int, n: int = 2, device: int = 0, env_id: str = None, **kwargs):
    """The batch as `n` independent sub-batches on ONE GPU, each a WarehouseVecEnv(output="torch") on a torch stream of its own —
    double-buffered sampling: `for pipe in pipes: with pipe as env: actions = policy(obs[pipe]); obs[pipe], *_ = env.step(actions)`.
    Nothing orders one sub-batch behind another, so the GPU runs the step of one beside the policy (or the step) of the other.
    Why bother: the workgroups of one step launch run their load / agent / store phases in lock-step and leave the memory system
    and the SIMDs idle in turns; two free-running half-size launches fill each other's gaps — measured on MI355X, per step of the
    whole batch: small-10ag x 16384 13.1 -> 10.7 us, large-16ag 18.1 -> 13.4, medium-6ag-hard x 16384 8.9 -> 6.7, small-4ag x 32768
    10.8 -> 7.0 (profiles/r04_two_pipelines.txt; bench.py `two_pipelines`).  Below ~2048 envs per sub-batch the launches are too
    short to overlap and the split costs (tiny-2ag x 4096: 4.3 -> 5.3).
    Returns a list of Pipeline (env, stream, lo, hi); results are identical to one env over the whole batch when sub-batch k is
    reset with `seed + lo` (Pipeline.reset does that)."""
    import torch

    from .vector_env import WarehouseVecEnv

    if n < 1 or num_envs % n:
        raise ValueError(f"num_envs {num_envs} must be a positive multiple of the number of pipelines {n}")
    kw = env_kwargs(env_id) if env_id else {}
    kw.update(kwargs)
    kw.pop("output", None)
    if "devices" in kw:
        raise ValueError("make_pipelines puts every sub-batch on ONE GPU: pass device=<ordinal>, not devices=")
    per = num_envs // n
    # Two launches in flight on one device: the raised wavefront priority each engine would give its own chain (rw_info.wave_priority)
    # takes issue slots from the OTHER pipeline's store phase — which is what fills the gap.  Measured (profiles/r06_pipelines_prio.txt, two
    # pipelines of 8192 envs, us per step of the whole batch, without / with): small-8ag 7.9 / 9.1, small-10ag 10.7 / 11.9, small-12ag
    # 11.9 / 12.9, medium-13ag 13.4 / 15.4, medium-6ag-hard 6.85 / 7.05; from 16384 envs per pipeline on the engine's own rule is as good
    # or better (small-8ag x 32768 13.3 / 12.0, small-4ag 7.2 / 7.1).  The caller's wave_priority= wins.
    if n > 1 and per <= 8192:
        kw.setdefault("wave_priority", False)
    pipes, rejected = [], []
    for k in range(n):
        # HIP maps streams onto a handful of hardware queues, and two streams that land on the same queue run one after the other
        # (measured: two streams created behind two busy ones shared a queue — 19.3 us per step instead of 10.7).  Nothing reports
        # the mapping, so each new stream is tried against the ones already chosen and replaced if it does not overlap with them.
        stream = torch.cuda.Stream(device=device)
        for _ in range(8):
            if all(streams_overlap(p.stream, stream) for p in pipes):
                break
            rejected.append(stream)  # (kept referenced: torch's stream pool then moves on to another one)
            stream = torch.cuda.Stream(device=device)
        else:
            import warnings

            warnings.warn("make_pipelines: no stream found that runs beside the other pipelines' streams; the sub-batches may serialise",
                          RuntimeWarning, stacklevel=2)
        with torch.cuda.stream(stream):  # (output="torch": the engine enqueues on torch's current stream at construction)
            env = WarehouseVecEnv(per, devices=[device], output="torch", **kw)
        pipes.append(Pipeline(env, stream, k * per, (k + 1) * per))
    return pipes


class _CapturedPipelines:
    """What capture_pipelines returns: `steps` (policy, step) rounds of every pipeline in ONE HIP graph, the pipelines side by side."""

    def __init__(self, graph, origin, pipes, keep, steps):
        self.graph, self.stream, self.pipes, self._keep, self.steps = graph, origin, pipes, keep, steps

    def replay(self):
        import torch

        for p in self.pipes:                       # behind whatever the pipelines have enqueued so far
            self.stream.wait_stream(p.stream)
        with torch.cuda.stream(self.stream):
            self.graph.replay()
        for p in self.pipes:                       # and their next eager work behind the replay
            p.stream.wait_stream(self.stream)
            p.env.engines[0].mark_views_stale()    # (the replayed steps ran without host code)


def capture_pipelines(pipes, policy, steps: int = 1, warmup: int = 2):
    """`steps` rounds of `actions = policy(obs, rewards, terminated); env.step(actions)` for EVERY pipeline of make_pipelines, captured in
    one HIP graph with one branch per pipeline (fork and join by events, nothing between the branches): on replay the GPU runs the
    policy of one sub-batch beside the step of the other with no host code in the loop.  `policy` as for WarehouseVecEnv.capture_loop
    (capturable, returns integer CUDA actions of its sub-batch); it is called `warmup` times per pipeline eagerly first.
    Returns an object with `.replay()`, `.graph` and `.stream` (the stream the graph is launched on)."""
    import torch

    dev = pipes[0].stream.device
    origin = torch.cuda.Stream(device=dev)
    views = []
    for p in pipes:
        env = p.env
        v = env._torch_views()
        views.append((env._obs_of(v), v["rewards"], v["terminated_bool"]))
        with torch.cuda.stream(p.stream):
            for _ in range(max(0, int(warmup))):
                policy(*views[-1])
            p.stream.synchronize()
        origin.wait_stream(p.stream)
    keep = []
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=origin):
        for p in pipes:
            p.stream.wait_stream(origin)           # fork: the pipeline's stream joins the capture
        for p, (obs, rew, term) in zip(pipes, views):
            env = p.env
            with torch.cuda.stream(p.stream):
                for _ in range(int(steps)):
                    a = env._device_actions(policy(obs, rew, term), env.num_envs, 0)
                    keep.append(a)                 # (allocated from the graph's private pool: alive as long as the graph is)
                    env.engines[0].step_device(a.data_ptr())
        for p in pipes:
            origin.wait_stream(p.stream)           # join
    return _CapturedPipelines(g, origin, list(pipes), keep, int(steps))


def streams_overlap(a, b, spin_cycles: int = 600_000) -> bool:
    """Do two torch streams of one device run side by side?  A spin kernel on `a` alone against one on each: concurrent streams take
    about the same wall time, streams that share a hardware queue twice as long (~1 ms in all)."""
    import time

    import torch

    if not hasattr(torch.cuda, "_sleep"):  # (no spin kernel to test with: assume they do)
        return True

    def timed(streams):
        for s_ in streams:
            s_.synchronize()
        t0 = time.perf_counter()
        for s_ in streams:
            with torch.cuda.stream(s_):
                torch.cuda._sleep(spin_cycles)
        for s_ in streams:
            s_.synchronize()
        return time.perf_counter() - t0

    timed([a, b])  # (first use of a stream: queue creation, not part of the comparison)
    alone = min(timed([a]) for _ in range(2))
    both = min(timed([a, b]) for _ in range(2))
    return both < 1.5 * alone


def register_gymnasium(override: bool = False) -> int:
    """Attach this engine as the `vector_entry_point` of every rware id (needs real gymnasium >= 1.0)."""
    import gymnasium as gym

    if getattr(gym, "IS_STANDIN", False):
        raise RuntimeError("refusing to register into the test stand-in")
    n = 0
    for env_id in all_ids():
        if env_id in gym.registry and not override:
            spec = gym.registry[env_id]
            if getattr(spec, "vector_entry_point", None) is None:
                spec.vector_entry_point = "rware_amd.vector_env:WarehouseVecEnv"
                n += 1
            continue
        gym.register(id=env_id, entry_point="rware.warehouse:Warehouse",
                     vector_entry_point="rware_amd.vector_env:WarehouseVecEnv", kwargs=env_kwargs(env_id))
        n += 1
    return n
