#!/usr/bin/env python
"""Per-workgroup phase timeline of one step (rw_debug_timeline): where does the time go?
Prints, per phase mark, the median/p10/p90 offset (us) from the earliest workgroup start, and
the distribution of workgroup start and end times across the grid.
The marks INSIDE the agent phases exist only in a library built with -DRW_TL_AG_MARKS
(`make -C robotic-warehouse_amd/csrc clean all CXXFLAGS="<the Makefile's flags> -DRW_TL_AG_MARKS"`): the product build
leaves them out, because even a switched-off mark is a scalar test + branch on the wavefront everybody waits for."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import rware_amd  # noqa: E402

MARKS = ["start", "zeroed", "dma_issued", "env_loaded", "loaded", "agents", "reset", "obs_bits", "obs_stored", "end"]


def main():
    env_id = sys.argv[1] if len(sys.argv) > 1 else "rware-small-4ag-v1"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
    E = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    T = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    kw = rware_amd.env_kwargs(env_id)
    if os.environ.get("TL_OBS_TYPE"):  # 2 IMAGE, 3 IMAGE_DICT
        kw["observation_type"] = int(os.environ["TL_OBS_TYPE"])
    if os.environ.get("TL_SENSOR_RANGE"):
        kw["sensor_range"] = int(os.environ["TL_SENSOR_RANGE"])
    env = rware_amd.WarehouseVecEnv(B, envs_per_workgroup=E, threads_per_workgroup=T, **kw)
    eng = env.engines[0]
    env.reset(seed=0)
    acts = torch.randint(0, 5, (32, B, kw["n_agents"]), dtype=torch.int32).cuda()
    for t in range(20):
        eng.step_device(acts[t].data_ptr())
    eng.sync()
    raw = eng.debug_timeline(acts[21].data_ptr()).astype(np.int64)
    tl = raw[:, : len(MARKS)]
    t0 = tl[:, 0].min()
    us = (tl - t0) / 100.0
    print(f"{env_id} B={B} E={eng.info.envs_per_workgroup} T={eng.info.threads_per_workgroup} wgs={tl.shape[0]}")
    print(f"{'mark':12s} {'p10':>8s} {'median':>8s} {'p90':>8s} {'max':>8s}   (us since first workgroup start)")
    for k, name in enumerate(MARKS):
        c = us[:, k]
        print(f"{name:12s} {np.percentile(c, 10):8.2f} {np.median(c):8.2f} {np.percentile(c, 90):8.2f} {c.max():8.2f}")
    d = np.diff(us, axis=1)
    print("per-workgroup phase durations (us): median / p90 / p99 / max")
    for k in range(len(MARKS) - 1):
        print(f"  {MARKS[k]:>10s} -> {MARKS[k + 1]:10s} {np.median(d[:, k]):7.2f} {np.percentile(d[:, k], 90):7.2f} "
              f"{np.percentile(d[:, k], 99):7.2f} {d[:, k].max():7.2f}")
    if raw.shape[1] >= 17 and raw[:, 12].max() > 0:  # marks inside the agent phases (register-exchange builds)
        names = ["loaded", "record read", "cells read", "winners", "applied", "goals", "agents"]
        cols = np.stack([raw[:, 4], raw[:, 12], raw[:, 13], raw[:, 14], raw[:, 15], raw[:, 16], raw[:, 5]], axis=1)
        dd = np.diff((cols - t0) / 100.0, axis=1)
        print("inside the agent phases (wavefront 0), us: median / p90 / p99 / max")
        for k in range(len(names) - 1):
            print(f"  {names[k]:>12s} -> {names[k + 1]:12s} {np.median(dd[:, k]):7.2f} {np.percentile(dd[:, k], 90):7.2f} "
                  f"{np.percentile(dd[:, k], 99):7.2f} {dd[:, k].max():7.2f}")
    life = us[:, -1] - us[:, 0]
    life = us[:, len(MARKS) - 1] - us[:, 0]
    last = np.argsort(us[:, -1])[-8:]
    print("the 8 workgroups that end last: id, start, end, lifetime, agents-phase")
    for b in last:
        print(f"  {b:5d} {us[b, 0]:6.2f} {us[b, -1]:6.2f} {life[b]:6.2f} {d[b, 4]:6.2f}")
    print(f"lifetime p50 {np.median(life):.2f} p90 {np.percentile(life, 90):.2f} p99 {np.percentile(life, 99):.2f} max {life.max():.2f}")
    print(f"workgroup lifetime median {np.median(us[:, -1] - us[:, 0]):.2f} us; kernel span {us[:, -1].max():.2f} us")


if __name__ == "__main__":
    main()
