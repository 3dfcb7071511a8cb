"""One line per task: the batch as ONE engine against rware_amd.make_pipelines(B, 2) (two sub-batches on streams of their own), us per
step of the WHOLE batch, per-step launches from device action tapes, a launcher thread per engine (un-profiled, wall clock).
    python profiles/tools/grid_pipelines.py [B [task[:sensor_range] ...]]   (no tasks: the papers' grid + the 9 .. 19-agent ids)"""
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import rware_amd  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
TASKS = sys.argv[2:]
grid = [f"rware-{s}-{n}ag{d}-v1" for s in ("tiny", "small", "medium") for n in (2, 4, 6, 8) for d in ("-easy", "", "-hard")]
extra = ["rware-large-2ag-v1", "rware-large-4ag-v1", "rware-large-6ag-v1", "rware-large-8ag-v1", "rware-small-9ag-v1", "rware-small-10ag-v1",
         "rware-tiny-11ag-v1", "rware-small-12ag-v1", "rware-medium-13ag-v1", "rware-small-14ag-v1", "rware-medium-15ag-hard-v1", "rware-small-16ag-v1",
         "rware-large-16ag-v1", "rware-small-17ag-v1", "rware-large-18ag-easy-v1", "rware-small-19ag-v1"]
K, TS = 1500, 64
print(f"B = {B} envs on one GPU; us per step of the whole batch; one engine -> two pipelines of {B // 2}")
print(f"{'task':30s} {'one engine':>11s} {'two pipelines':>14s} {'change':>8s} {'G agent-steps/s':>16s}")
for env_id in (TASKS or grid + extra):
    sr = 0
    if ":" in env_id:
        env_id, sr = env_id.split(":")[0], int(env_id.split(":")[1])
    kw = rware_amd.env_kwargs(env_id)
    if sr:
        kw["sensor_range"] = sr
    N = kw["n_agents"]
    acts = torch.from_numpy(np.random.default_rng(1).integers(0, 5, size=(TS, B, N), dtype=np.int32)).cuda()
    res = []
    for M in (1, 2):
        if M == 1:
            envs = [rware_amd.WarehouseVecEnv(B, output="torch", **kw)]
            envs[0].reset(seed=0)
            bounds = [(0, B)]
        else:
            pipes = rware_amd.make_pipelines(B, 2, **kw)
            for p in pipes:
                p.reset(seed=0)
            envs, bounds = [p.env for p in pipes], [(p.lo, p.hi) for p in pipes]
        engs = [e.engines[0] for e in envs]
        tapes = [acts[:, lo:hi].contiguous() for lo, hi in bounds]
        torch.cuda.synchronize()

        def run(k, n):
            engs[k].step_tape_device(tapes[k].data_ptr(), TS, 0, n)
            engs[k].sync()
        for k in range(M):
            run(k, 200)
        best = None
        for _ in range(3):
            th = [threading.Thread(target=run, args=(k, K)) for k in range(M)]
            t0 = time.perf_counter()
            for t in th:
                t.start()
            for t in th:
                t.join()
            us = (time.perf_counter() - t0) / K * 1e6
            best = us if best is None else min(best, us)
        res.append(best)
        for e in envs:
            e.close()
    print(f"{env_id:30s} {res[0]:11.3f} {res[1]:14.3f} {res[1] / res[0] - 1:+8.1%} {B * N / min(res) / 1e3:16.2f}", flush=True)
