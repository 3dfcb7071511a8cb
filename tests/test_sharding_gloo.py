"""N>1 path on CPU: world_size-2 `gloo` processes.  Each rank owns a contiguous env shard seeded by
GLOBAL env index (no data-path collective — envs are independent), and the only communication is
bench.py's barrier + MAX-over-ranks of the elapsed time.  The shard engines are the PRODUCT — the
engine sources built for host threads (tests/emu), driven through `WarehouseVecEnv` and seeded with
`rware_amd.shard_seeds(rank, B)` exactly as bench.py does — and the claim under test is that the
concatenation of the shards equals the UNSHARDED oracle run, i.e. results do not depend on the GPU count."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B_PER_RANK, N, T = 6, 4, 40


def _kw():
    sys.path.insert(0, ROOT)
    import rware_amd
    kw = rware_amd.env_kwargs("rware-small-4ag-v1")
    kw["reward_type"] = kw["reward_type"].value
    kw["max_steps"] = 17
    return kw


def _actions(world):
    return np.random.default_rng(4).integers(0, 5, size=(T, world * B_PER_RANK, N), dtype=np.int32)


def _run_shard(rank, world):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import rware_amd
    from engine_backend import EMU_LIB
    env = rware_amd.WarehouseVecEnv(B_PER_RANK, library=EMU_LIB, envs_per_workgroup=4, threads_per_workgroup=64, **_kw())
    env.engines[0].reset(seeds=rware_amd.shard_seeds(rank, B_PER_RANK))   # the very line bench.py seeds its shard with
    acts = _actions(world)[:, rank * B_PER_RANK:(rank + 1) * B_PER_RANK]
    out = [env.step(a) for a in acts]
    st = env.get_state()
    env.close()
    return np.stack([o[0] for o in out]), np.stack([o[1] for o in out]), st


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dist.barrier()
    obs, rew, st = _run_shard(rank, world)
    elapsed = torch.tensor([1.0 + rank], dtype=torch.float64)   # bench.py: MAX over ranks
    dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    steps = torch.tensor([float(obs.shape[0] * obs.shape[1] * N)], dtype=torch.float64)
    dist.all_reduce(steps, op=dist.ReduceOp.SUM)
    dist.barrier()
    q.put((rank, obs, rew, st["rng"], float(elapsed), float(steps)))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_shards_equal_the_unsharded_batch():
    world = 2
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from engine_backend import build_emu
    build_emu()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from rware_oracle import OracleVecEnv
    full = OracleVecEnv(world * B_PER_RANK, **_kw())
    full.reset(seed=0)                                 # env i <- SeedSequence(i)
    outs = [full.step_autoreset(a, "next_step") for a in _actions(world)]
    obs_full = np.stack([o[0] for o in outs])
    rew_full = np.stack([o[1] for o in outs])
    assert np.array_equal(np.concatenate([g[1] for g in got], axis=1), obs_full)
    assert np.array_equal(np.concatenate([g[2] for g in got], axis=1), rew_full)
    assert np.array_equal(np.concatenate([g[3] for g in got], axis=0), full.get_state()["rng"])
    assert all(g[4] == 2.0 for g in got)               # MAX over ranks
    assert all(g[5] == world * B_PER_RANK * T * N for g in got)
