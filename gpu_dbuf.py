# experiment (git-ignored): the batch as K sub-batches on K engines (= K streams) stepping concurrently
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import rware_amd
B, N, STEPS, TS = 16384, 4, 4000, 64
kw = rware_amd.env_kwargs("rware-small-4ag-v1")
for K in (1, 2, 4):
    b = B // K
    envs = [rware_amd.WarehouseVecEnv(b, **kw) for _ in range(K)]
    for i, e in enumerate(envs):
        e.engines[0].reset(seeds=rware_amd.shard_seeds(i, b))
    tapes = [torch.randint(0, 5, (TS, b, N), dtype=torch.int32, device="cuda") for _ in range(K)]
    for e, t in zip(envs, tapes):
        e.engines[0].step_tape_device(t.data_ptr(), TS, 0, 200)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    CH = 100
    for c in range(STEPS // CH):
        for e, t in zip(envs, tapes):
            e.engines[0].step_tape_device(t.data_ptr(), TS, (c * CH) % TS, CH)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"K={K} sub-batches of {b}: {dt / STEPS * 1e6:.3f} us per full-batch step, {B * N * STEPS / dt / 1e9:.2f} G agent-steps/s")
    for e in envs:
        e.close()
