"""Does a consumer that reads the observations right behind every step care how they were stored?
Closed loop on one stream, no sync inside: a = argmax(obs.view(B*N, L) @ W) ; env.step(a) — a one-layer "policy" that reads
the whole observation tensor every step — with cached and with non-temporal observation stores (obs_stores=)."""
import sys
import time

import torch

sys.path.insert(0, ".")
import rware_amd  # noqa: E402

B, STEPS = 16384, 2000
kw = rware_amd.env_kwargs("rware-small-4ag-v1")
for pol in ("cached", "stream", "cached", "stream"):
    env = rware_amd.WarehouseVecEnv(B, output="torch", obs_stores=pol, **kw)
    obs, _ = env.reset(seed=0)
    W = torch.randn(obs.shape[-1], 5, device="cuda")
    def policy(o):
        return (o.view(-1, o.shape[-1]) @ W).argmax(-1).to(torch.int32).view(B, -1)
    for _ in range(200):
        obs, rew, term, trunc, _ = env.step(policy(obs))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(STEPS):
        obs, rew, term, trunc, _ = env.step(policy(obs))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / STEPS
    env.sync()
    print(f"obs_stores={pol:7s}: {dt * 1e6:7.2f} us per (policy + step) round, {B * 4 / dt / 1e9:.2f} G agent-steps/s", flush=True)
    env.close()

# the same policy + step, captured in one HIP graph (WarehouseVecEnv.capture_loop): K rounds per replay
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    env = rware_amd.WarehouseVecEnv(B, output="torch", **kw)
    obs, _ = env.reset(seed=0)
    W = torch.randn(obs.shape[-1], 5, device="cuda")
    K = 50
    loop = env.capture_loop(lambda o, r, d: (o.view(-1, o.shape[-1]) @ W).argmax(-1).view(B, -1), steps=K)
    for _ in range(4):
        loop.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(STEPS // K):
        loop.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (STEPS // K * K)
    env.sync()
    print(f"captured in a HIP graph ({K} rounds per replay): {dt * 1e6:7.2f} us per (policy + step) round, {B * 4 / dt / 1e9:.2f} G agent-steps/s", flush=True)
    env.close()
