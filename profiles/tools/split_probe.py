"""Would the headline batch run faster as M independent sub-batches on M streams (one engine each, own launcher thread)?
    python profiles/tools/split_probe.py [env_id] [B] [M ...]
Each engine steps its B / M envs K times from a device action tape (rw_step_tape_device: a C loop of launches on the engine's own
stream, GIL released); wall clock from the first enqueue to the last sync.  M = 1 is the ordinary engine."""
import os
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import rware_amd  # noqa: E402

env_id = sys.argv[1] if len(sys.argv) > 1 else "rware-small-4ag-v1"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
Ms = [int(x) for x in sys.argv[3:]] or [1, 2, 4]
kw = rware_amd.env_kwargs(env_id)
if os.environ.get("SPLIT_SENSOR_RANGE"):   # (BASELINE config 5: rware-large-16ag-v1 with sensor_range 2)
    kw["sensor_range"] = int(os.environ["SPLIT_SENSOR_RANGE"])
N = kw["n_agents"]
K, T = 4000, 64
for M in Ms:
    env = rware_amd.WarehouseVecEnv(B, devices=[0] * M, **kw) if M > 1 else rware_amd.WarehouseVecEnv(B, **kw)
    env.reset(seed=1)
    engs = env.engines
    tapes = [torch.from_numpy(np.random.default_rng(k).integers(0, 5, size=(T, B // M, N), dtype=np.int32)).cuda() for k in range(M)]
    def run(k, n):
        engs[k].step_tape_device(tapes[k].data_ptr(), T, 0, n)
        engs[k].sync()
    for k in range(M):
        run(k, 200)
    best = None
    for rep in range(3):
        th = [threading.Thread(target=run, args=(k, K)) for k in range(M)]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        us = (time.perf_counter() - t0) / K * 1e6
        best = us if best is None else min(best, us)
    i = engs[0].info
    print(f"{env_id} B={B} as {M} x {B // M} envs: {best:7.3f} us per step of the whole batch, {B * N / best / 1e3:6.2f} G agent-steps/s "
          f"(E {int(i.envs_per_workgroup)}, {int(i.n_workgroups)} workgroups per engine)", flush=True)
    env.close()
