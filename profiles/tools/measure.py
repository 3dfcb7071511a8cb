"""us per step of explicit (task, batch, geometry, observation-store) combinations, un-profiled, HIP events on the launches.
    python profiles/tools/measure.py <env_id>:<B>[:<E>[:<stores>[:<sensor_range>[:<jit>[:<fused>[:<pipe>[:<threads per workgroup>]]]]]]] ...
E = 0: the engine's own geometry; stores = auto | cached | stream; jit = auto | off | force (run-time specialisation);
fused = n: the fused rollout instead, n steps per launch (rw_step_many_device, wall clock around 32 launches);
pipe = auto | off | on: the chunk-pipelined persistent per-step kernel (RWARE_PIPE_E / RWARE_PIPE_WGS_PER_CU pick its geometry).
<env_id> may also be one of the unregistered shapes below (constructor arguments, not ids).  Per-step launches from a device action tape
(rw_step_tape_device_timed), uniform random actions, next_step autoreset.  One line per spec."""
import os
import sys

import numpy as np

os.environ.setdefault("RWARE_HOOKS", "1")  # (this tool drives the library's A/B hooks: csrc/rware_hooks.h)
import torch

sys.path.insert(0, ".")
import rware_amd  # noqa: E402

KIND = {0: "generic", 1: "exact", 2: "agent-count-static", 3: "size-static"}
CUSTOM = {
    # the layout string of the reference's README / tests (rware/warehouse.py:328-350), 3 agents
    "layoutstr-3ag": dict(layout="\n".join(["X.....X", "X.....X", "X.xxx.X", "X.xxx.X", "X.....X", "X.....X", "..g.g.."]), n_agents=3, request_queue_size=3),
    "small-4ag-colheight5": dict(rware_amd.env_kwargs("rware-small-4ag-v1"), column_height=5),
    "sr5-12ag-colheight5": dict(shelf_columns=3, shelf_rows=2, column_height=5, n_agents=12, request_queue_size=12, sensor_range=5),
    "small-24ag": dict(rware_amd.env_kwargs("rware-small-4ag-v1"), n_agents=24, request_queue_size=24),
}
for spec in sys.argv[1:]:
    f = spec.split(":")
    env_id, B = f[0], int(f[1])
    E = int(f[2]) if len(f) > 2 else 0
    stores = f[3] if len(f) > 3 and f[3] != "auto" else None
    sr = int(f[4]) if len(f) > 4 and f[4] else 0
    jit = {"auto": None, "off": False, "force": True}[f[5]] if len(f) > 5 and f[5] else None
    fused = int(f[6]) if len(f) > 6 and f[6] else 0
    pipe = {"auto": None, "off": False, "on": True}[f[7]] if len(f) > 7 and f[7] else None
    T_wg = int(f[8]) if len(f) > 8 and f[8] else 256
    kw = dict(CUSTOM[env_id]) if env_id in CUSTOM else rware_amd.env_kwargs(env_id)
    if sr:
        kw["sensor_range"] = sr
    N = kw["n_agents"]
    try:
        env = rware_amd.WarehouseVecEnv(B, envs_per_workgroup=E, threads_per_workgroup=T_wg if E else 0, obs_stores=stores, jit=jit, pipe=pipe, **kw)
    except Exception as exc:  # noqa: BLE001
        print(f"{spec:44s} FAILED {exc}")
        continue
    eng = env.engines[0]
    eng.reset(seeds=rware_amd.shard_seeds(0, B))
    T = 32 if B * N > 1 << 20 else 64
    T = max(T, fused)
    tape = torch.from_numpy(np.random.default_rng(1).integers(0, 5, size=(T, B, N), dtype=np.int32)).cuda()
    K = 2000 if B * N <= 1 << 18 else 400
    eng.step_tape_device_timed(tape.data_ptr(), T, 0, max(K // 8, 50), 0, 1)
    torch.cuda.synchronize()
    best = None
    for rep in range(2 if fused else 0):
        import time
        eng.step_many_device(tape.data_ptr(), fused)
        eng.sync()
        t0 = time.perf_counter()
        for _ in range(32):
            eng.step_many_device(tape.data_ptr(), fused)
        eng.sync()
        us = (time.perf_counter() - t0) / (32 * fused) * 1e6
        best = us if best is None else min(best, us)
    for rep in range(0 if fused else 2):
        eng.step_tape_device_timed(tape.data_ptr(), T, 0, K, 0, 1)
        torch.cuda.synchronize()
        us = eng.event_elapsed_ms(0, 1) / K * 1e3
        best = us if best is None else min(best, us)
    eng.sync()
    i = eng.info
    eb = int(i.engine_bytes_per_env_step) * B
    geom = f"pipe E {int(i.pipe_envs_per_workgroup):2d} x {int(i.pipe_workgroups):4d} wgs" if int(i.pipe_workgroups) else f"E {int(i.envs_per_workgroup):2d}" + (f" T {int(i.threads_per_workgroup)}" if int(i.threads_per_workgroup) != 256 else "")
    print(f"{spec:44s} {KIND[int(i.build_kind)] + (' (jit)' if int(i.jit) > 0 else ''):18s} {geom} {'nt' if int(i.obs_stores_stream) else 'cached':6s} {'prio' if int(i.wave_priority) else '    '} "
          f"{best:8.3f} us/step {B * N / best / 1e3:7.2f} G a-s/s  engine {eb / 1e6:7.1f} MB -> {eb / best / 1e6:5.2f} TB/s = {eb / best / 8e6:4.2f} of peak", flush=True)
    env.close()
