"""PCIe-inclusive rate of the host-array API: WarehouseVecEnv(output="numpy").step(numpy actions) -> numpy results, one call per
step (actions staged through the engine's pinned buffer, results read back with rw_read_outputs: one synchronisation)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import rware_amd
kw = rware_amd.env_kwargs("rware-small-4ag-v1")
for B in (1024, 16384):
    env = rware_amd.WarehouseVecEnv(B, **kw)
    env.reset(seed=0)
    acts = np.random.default_rng(0).integers(0, 5, size=(32, B, 4), dtype=np.int32)
    for t in range(20):
        env.step(acts[t % 32])
    n = 200
    t0 = time.perf_counter()
    for t in range(n):
        obs, rew, term, trunc, _ = env.step(acts[t % 32])
    dt = (time.perf_counter() - t0) / n
    print(f"numpy in/out  B={B}: {dt*1e6:9.1f} us/step  {B*4/dt/1e6:8.1f} M agent-steps/s  obs {obs.nbytes/1e6:.1f} MB -> {obs.nbytes/dt/1e9:.1f} GB/s over PCIe")
    env.close()
