# experiment (git-ignored): per-step launches replayed from a HIP graph vs issued one by one
import time, sys
import numpy as np, torch
sys.path.insert(0, ".")
import rware_amd
B, K = 16384, 100
kw = rware_amd.env_kwargs("rware-small-4ag-v1")
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    env = rware_amd.WarehouseVecEnv(B, output="torch", **kw)
    eng = env.engines[0]
    env.reset(seed=0)
    tape = torch.randint(0, 5, (K, B, 4), dtype=torch.int32, device="cuda")
    base, stride = tape.data_ptr(), B * 4 * 4
    for t in range(K):
        eng.step_device(base + t * stride)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for r in range(20):
        eng.step_tape_device(base, K, 0, K)
    torch.cuda.synchronize()
    print("plain  us/step %.3f" % ((time.perf_counter() - t0) / (20 * K) * 1e6))
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        eng.step_tape_device(base, K, 0, K)
    torch.cuda.synchronize()
    for r in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for r in range(20):
        g.replay()
    torch.cuda.synchronize()
    print("graph  us/step %.3f" % ((time.perf_counter() - t0) / (20 * K) * 1e6))
