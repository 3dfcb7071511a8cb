#!/usr/bin/env python
"""Headline benchmark: agent-steps/s of the HIP step engine on rware-small-4ag, batch 16384 per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path (Warehouse.step + FLATTENED obs, rware/warehouse.py:804-946)
over the whole env batch resident in HBM: one `rw_step_device` call, actions read from a
device-resident tape, NEXT_STEP autoreset on (all envs reset on-device every 500 steps, inside
the timed region).  Multi-GPU: one process per GPU, each with its own 16384-env shard (weak
scaling), seeds offset by the global env index, NO data-path collective (envs are independent;
SURVEY.md §8(e)); torch.distributed is used only for the barrier and the MAX over ranks.

The printed JSON line carries `roofline` (algorithmic bytes per launch / HIP-event time per launch
on the engine's stream vs 8 TB/s HBM) and, at N=1, `cpu_baseline` (the C oracle — a port of the
reference step — timed on one host core over a bounded sample of the same workload).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

ENV_ID = "rware-small-4ag-v1"
BATCH_PER_GPU = 16384
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy ceiling)
TAPE_STEPS = int(os.environ.get("RWARE_BENCH_TAPE_STEPS", "256"))


def cpu_baseline(seconds: float = 12.0):
    """Oracle (C port of the reference step + obs) on ONE host core, bounded sample."""
    import rware_amd
    from rware_oracle import OracleVecEnv

    kw = rware_amd.env_kwargs(ENV_ID)
    kw["reward_type"] = kw["reward_type"].value
    b = 512
    env = OracleVecEnv(b, **kw)
    env.reset(seed=0)
    rng = np.random.default_rng(12345)
    acts = rng.integers(0, 5, size=(64, b, kw["n_agents"]), dtype=np.int32)
    for t in range(8):
        env.step_autoreset(acts[t % 64], "next_step")
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for t in range(16):
            env.step_autoreset(acts[(n + t) % 64], "next_step")
        n += 16
    dt = time.perf_counter() - t0
    return {
        "value": b * kw["n_agents"] * n / dt,
        "unit": "agent-steps/s",
        "cores": 1,
        "kind": "port",
        "sample": f"{ENV_ID}, {b} envs x {n} steps (~{dt:.0f} s), uniform random actions, step+obs, next_step autoreset, "
                  "oracle/rware_oracle.c single thread",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20000)
    ap.add_argument("--warmup", type=int, default=500)
    ap.add_argument("--batch", type=int, default=BATCH_PER_GPU, help="envs per GPU (headline: 16384)")
    ap.add_argument("--env-id", default=ENV_ID)
    ap.add_argument("--sensor-range", type=int, default=0, help="override sensor_range (BASELINE config 5 uses 2)")
    ap.add_argument("--observation-type", type=int, default=1, help="1 FLATTENED (headline), 2 IMAGE, 3 IMAGE_DICT")
    ap.add_argument("--envs-per-wg", type=int, default=0)
    ap.add_argument("--threads-per-wg", type=int, default=0)
    ap.add_argument("--many", type=int, default=0,
                    help="fused rollout: submit steps in chunks of this many through rw_step_many_device (one launch per chunk, "
                         "env chunk resident in LDS across the steps; open-loop)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fused-extra", action="store_true", help="skip the extra fused-rollout measurement (profiling runs)")
    args = ap.parse_args()

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this driver stack (RCCL across processes)
    import torch

    import rware_amd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch N>1 with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the engine has no CPU fallback")
    # Test hooks (tests/test_gpu_parity.py runs the N > 1 path on a 1-GPU box): RWARE_BENCH_BACKEND=gloo replaces
    # RCCL for the barrier / MAX-over-ranks, RWARE_BENCH_SHARE_GPU=1 lets several ranks share one device.
    backend = os.environ.get("RWARE_BENCH_BACKEND", "nccl")
    if os.environ.get("RWARE_BENCH_SHARE_GPU") == "1":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)

    kw = rware_amd.env_kwargs(args.env_id)
    if args.sensor_range:
        kw["sensor_range"] = args.sensor_range
    if args.observation_type != 1:
        kw["observation_type"] = args.observation_type
    B, N = args.batch, kw["n_agents"]
    env = rware_amd.WarehouseVecEnv(B, devices=[local_rank], envs_per_workgroup=args.envs_per_wg,
                                    threads_per_workgroup=args.threads_per_wg, **kw)
    eng = env.engines[0]
    info = eng.info
    # env i of rank r is global env r*B + i -> SeedSequence(r*B + i): results independent of the GPU count
    eng.reset(seeds=np.arange(rank * B, (rank + 1) * B, dtype=np.uint64))
    tape = torch.from_numpy(
        np.random.default_rng(12345 + rank).integers(0, 5, size=(TAPE_STEPS, B, N), dtype=np.int32)
    ).to(f"cuda:{local_rank}")
    base, stride = tape.data_ptr(), B * N * 4

    def run(n, t_start):
        if args.many > 0:
            t = t_start
            while t < t_start + n:
                c = min(args.many, t_start + n - t, TAPE_STEPS - (t % TAPE_STEPS))
                eng.step_many_device(base + (t % TAPE_STEPS) * stride, c)
                t += c
        else:
            for t in range(t_start, t_start + n):
                eng.step_device(base + (t % TAPE_STEPS) * stride)

    def barrier():
        if dist is not None:
            dist.barrier()

    run(args.warmup, 0)
    eng.sync()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.event_record(0)
    run(args.steps, args.warmup)
    eng.event_record(1)
    eng.sync()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = eng.event_elapsed_ms(0, 1) / max(args.steps, 1)  # avg per launch on the engine's stream

    # Extra (reported beside the headline, never as `value`): the same K steps through the fused rollout
    # API, rw_step_many_device — one launch per 64 steps, env chunk resident in LDS (open-loop).
    fused_elapsed = None
    if not args.many and not args.no_fused_extra:
        chunk = 64
        def run_fused(n, t_start):
            t = t_start
            while t < t_start + n:
                c = min(chunk, t_start + n - t, TAPE_STEPS - (t % TAPE_STEPS))
                eng.step_many_device(base + (t % TAPE_STEPS) * stride, c)
                t += c
        run_fused(min(args.warmup, 64), 0)
        eng.sync()
        torch.cuda.synchronize()
        barrier()
        tf = time.perf_counter()
        run_fused(args.steps, args.warmup)
        eng.sync()
        torch.cuda.synchronize()
        barrier()
        fused_elapsed = time.perf_counter() - tf

    if dist is not None:
        tt = torch.tensor([elapsed, kernel_ms, fused_elapsed or 0.0], dtype=torch.float64,
                          device=f"cuda:{local_rank}" if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed, kernel_ms = float(tt[0]), float(tt[1])
        fused_elapsed = float(tt[2]) or None

    if rank == 0:
        a_bytes = int(info.algorithmic_bytes_per_env_step)
        per_launch = a_bytes * B
        achieved = per_launch / (kernel_ms * 1e-3) / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")  # written from the rocprofv3 --pmc passes
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get(f"{args.env_id}:{B}")
            except Exception:
                traffic = None
        out = {
            "metric": "agent-steps/sec (agents*envs*steps/s)",
            "value": world * B * N * args.steps / elapsed,
            "unit": "agent-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int32",
            "data": "synthetic",
            "config": {
                "workload": f"{args.env_id} batch={B} envs per GPU, uniform random actions from a device tape, "
                            "step+FLATTENED obs, on-device next_step autoreset every 500 steps",
                "envs_per_gpu": B, "n_agents": N, "obs_length": int(info.obs_length),
                "grid": [int(info.grid_h), int(info.grid_w)],
                "parallelism": f"env-shard x{world} (no collective)",
                "submit": f"rw_step_many_device x{args.many} (fused rollout, one launch per chunk)" if args.many
                          else "rw_step_device per step (one launch per step, closed-loop capable)",
                "envs_per_workgroup": int(info.envs_per_workgroup), "threads_per_workgroup": int(info.threads_per_workgroup),
                "kernel_specialised": bool(info.specialised),
                "device": info.device_name.decode(), "arch": info.arch_name.decode(),
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                "kernel": "rw::rware_step_kernel", "kernel_ms_per_launch": kernel_ms,
                "algorithmic_bytes_per_launch": per_launch,
            },
        }
        if fused_elapsed:
            out["fused_rollout"] = {
                "value": world * B * N * args.steps / fused_elapsed, "unit": "agent-steps/s",
                "ms_per_step": fused_elapsed / args.steps * 1e3,
                "submit": "rw_step_many_device x64: one launch per 64 steps, env chunk resident in LDS across steps "
                          "(open-loop rollout from the same device action tape; identical results)",
            }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    env.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
