# helper: is a cache-exceeding batch faster as two half-batch engines stepped one after the other, or on streams of their own?
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import rware_amd
kw = rware_amd.env_kwargs("rware-small-4ag-v1")
TAPE = 16
def run(parts, B, same_stream, steps=300):
    b = B // parts
    envs = []
    if same_stream:
        for i in range(parts):
            envs.append(rware_amd.WarehouseVecEnv(b, output="torch", **kw))
    else:
        for i in range(parts):
            envs.append(rware_amd.WarehouseVecEnv(b, **kw))
    engs = [e.engines[0] for e in envs]
    for i, g in enumerate(engs):
        g.reset(seeds=rware_amd.shard_seeds(i, b))
    tape = torch.randint(0, 5, (TAPE, b, 4), dtype=torch.int32, device="cuda")
    base = tape.data_ptr()
    def go(n, t0):
        for t in range(t0, t0 + n):
            for g in engs:
                g.step_tape_device(base, TAPE, t % TAPE, 1)
    go(30, 0); torch.cuda.synchronize()
    t0 = time.perf_counter(); go(steps, 30); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    for e in envs: e.close()
    return dt / steps * 1e6
for B in (262144, 524288):
    for parts, same in ((1, True), (2, True), (2, False), (4, True), (4, False)):
        try:
            print(f"B={B} parts={parts} {'one stream' if same else 'own streams'}: {run(parts, B, same):8.2f} us per full step", flush=True)
        except Exception as e:
            print("failed", B, parts, same, repr(e)[:200], flush=True)
