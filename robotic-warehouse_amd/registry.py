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


def env_kwargs(env_id: str) -> dict:
    m = _ID.match(env_id)
    if not m or not (1 <= int(m.group(2)) <= 19):
        raise KeyError(f"unknown rware id {env_id!r}")
    size, agents, diff = m.group(1), int(m.group(2)), m.group(3) or ""
    return {
        "column_height": 8,
        "shelf_rows": _SIZES[size][0],
        "shelf_columns": _SIZES[size][1],
        "n_agents": agents,
        "msg_bits": 0,
        "sensor_range": 1,
        "request_queue_size": int(agents * _DIFFICULTY[diff]),
        "max_inactivity_steps": None,
        "max_steps": 500,
        "reward_type": RewardType.INDIVIDUAL,
    }


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
