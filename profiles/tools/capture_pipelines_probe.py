"""Closed loop with a policy in the loop, us per round of the WHOLE batch: eager one env / capture_loop (one env, one graph) /
make_pipelines eager from one host thread / capture_pipelines (two branches in one graph).  The policy: one linear layer on the
flattened observation + argmax (the smallest thing that still reads every observation and writes every action).
    python profiles/tools/capture_pipelines_probe.py [env_id] [B]"""
import sys
import time

import torch

sys.path.insert(0, ".")
import rware_amd  # noqa: E402

env_id = sys.argv[1] if len(sys.argv) > 1 else "rware-small-4ag-v1"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
kw = rware_amd.env_kwargs(env_id)
N = kw["n_agents"]
L = rware_amd.obs_length(kw.get("sensor_range", 1)) if hasattr(rware_amd, "obs_length") else 71
torch.manual_seed(0)
W = None


def policy(obs, rew, term):
    global W
    if W is None:
        W = torch.randn(obs.shape[-1], 5, device=obs.device) * 0.1
    return (obs @ W).argmax(-1).to(torch.int32)


ROUNDS, K = 64, 20   # K replays of ROUNDS rounds
res = {}
env = rware_amd.WarehouseVecEnv(B, output="torch", **kw)
obs = env.reset(seed=0)[0]
rew = torch.zeros((B, N), device="cuda"); term = torch.zeros((B,), dtype=torch.bool, device="cuda")
for _ in range(50):
    obs, rew, term, _, _ = env.step(policy(obs, rew, term))
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(ROUNDS * K):
    obs, rew, term, _, _ = env.step(policy(obs, rew, term))
torch.cuda.synchronize(); res["eager, one env"] = (time.perf_counter() - t0) / (ROUNDS * K) * 1e6
loop = env.capture_loop(policy, steps=ROUNDS)
loop.replay(); torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(K):
    loop.replay()
torch.cuda.synchronize(); res["capture_loop, one env"] = (time.perf_counter() - t0) / (ROUNDS * K) * 1e6
env.close()

pipes = rware_amd.make_pipelines(B, 2, **kw)
st = [(p.reset(seed=0)[0], None, None) for p in pipes]
v = []
for p in pipes:
    vv = p.env._torch_views()
    v.append([p.env._obs_of(vv), vv["rewards"], vv["terminated_bool"]])
for _ in range(50):
    for k, p in enumerate(pipes):
        with p as e:
            e.step(policy(*v[k]))
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(ROUNDS * K):
    for k, p in enumerate(pipes):
        with p as e:
            e.step(policy(*v[k]))
torch.cuda.synchronize(); res["eager, two pipelines (one host thread)"] = (time.perf_counter() - t0) / (ROUNDS * K) * 1e6
cp = rware_amd.capture_pipelines(pipes, policy, steps=ROUNDS)
cp.replay(); torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(K):
    cp.replay()
torch.cuda.synchronize(); res["capture_pipelines, two branches in one graph"] = (time.perf_counter() - t0) / (ROUNDS * K) * 1e6
for p in pipes:
    p.env.close()
print(f"{env_id} x {B} envs, one-layer policy + argmax in the loop; us per round of the whole batch")
for k, x in res.items():
    print(f"  {k:48s} {x:8.2f} us   {B * N / x / 1e3:6.2f} G agent-steps/s")
