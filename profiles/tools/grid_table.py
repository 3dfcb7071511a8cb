"""One line per task: which kernel build runs it and how fast, at B envs on one GPU (un-profiled; HIP events on the launches).
    python profiles/tools/grid_table.py [B]            # the papers' grid (36 ids) + odd agent counts + the large warehouse + 9..19 agents
Per-step launches from a device action tape (rw_step_tape_device_timed), uniform random actions, next_step autoreset."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import rware_amd  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
KIND = {0: "generic", 1: "exact", 2: "agent-count-static", 3: "size-static"}
grid = [f"rware-{s}-{n}ag{d}-v1" for s in ("tiny", "small", "medium") for n in (2, 4, 6, 8) for d in ("-easy", "", "-hard")]
extra = ["rware-small-1ag-v1", "rware-small-3ag-v1", "rware-small-5ag-v1", "rware-small-7ag-v1", "rware-tiny-3ag-hard-v1", "rware-medium-5ag-easy-v1",
         "rware-large-2ag-v1", "rware-large-4ag-v1", "rware-large-6ag-v1", "rware-large-8ag-v1",
         # 9 .. 19 agents: agent-count-static builds, agent phases in registers (round 4)
         "rware-small-9ag-v1", "rware-small-10ag-v1", "rware-tiny-11ag-v1", "rware-small-12ag-v1", "rware-medium-13ag-v1", "rware-small-14ag-v1",
         "rware-medium-15ag-hard-v1", "rware-small-16ag-v1", "rware-large-16ag-v1", "rware-small-17ag-v1", "rware-large-18ag-easy-v1",
         "rware-small-19ag-v1"]
print(f"B = {B} envs per GPU; us per step of the whole batch; G agent-steps/s")
print(f"{'task':30s} {'build':20s} {'E':>3s} {'obs stores':>12s} {'us/step':>9s} {'G a-s/s':>9s}")
for env_id in grid + extra:
    kw = rware_amd.env_kwargs(env_id)
    N = kw["n_agents"]
    env = rware_amd.WarehouseVecEnv(B, **kw)
    eng = env.engines[0]
    eng.reset(seeds=rware_amd.shard_seeds(0, B))
    T = 64
    tape = torch.from_numpy(np.random.default_rng(1).integers(0, 5, size=(T, B, N), dtype=np.int32)).cuda()
    eng.step_tape_device_timed(tape.data_ptr(), T, 0, 300, 0, 1)
    torch.cuda.synchronize()
    K = 2000
    eng.step_tape_device_timed(tape.data_ptr(), T, 300 % T, K, 0, 1)
    torch.cuda.synchronize()
    us = eng.event_elapsed_ms(0, 1) / K * 1e3
    eng.sync()
    i = eng.info
    print(f"{env_id:30s} {KIND[int(i.build_kind)]:20s} {int(i.envs_per_workgroup):3d} {'non-temporal' if int(i.obs_stores_stream) else 'cached':>12s} "
          f"{us:9.3f} {B * N / us / 1e3:9.2f}", flush=True)
    env.close()
