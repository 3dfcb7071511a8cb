"""Where do the microseconds of a SHORT timed region go?  (bench.py --steps 20: 20 launches = 150 us of GPU work.)

Times, for K launches submitted by rw_step_tape_device after a device-wide synchronise:
  submit   host time until the native launch loop returned
  total    host time until the closing synchronise returned
for several ways of bracketing the region.  Prints medians over TRIALS regions.  MEASUREMENT TOOL, not product code.
"""
import ctypes as C
import os
import statistics
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import rware_amd  # noqa: E402

K = int(os.environ.get("K20_STEPS", 20))
TRIALS = int(os.environ.get("K20_TRIALS", 200))
B = 16384
TAPE = 256

hip = C.CDLL("libamdhip64.so")


def main():
    kw = rware_amd.env_kwargs("rware-small-4ag-v1")
    N = kw["n_agents"]
    env = rware_amd.WarehouseVecEnv(B, devices=[0], **kw)
    eng = env.engines[0]
    eng.reset(seeds=rware_amd.shard_seeds(0, B))
    acts = np.random.default_rng(12345).integers(0, 5, size=(TAPE, B, N), dtype=np.int32)
    tape = torch.from_numpy(acts).cuda()
    base = tape.data_ptr()
    eng.step_tape_device(base, TAPE, 0, 200)
    torch.cuda.synchronize()

    def region(kind, t_first):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if kind in ("events+2sync", "events+1sync"):
            eng.event_record(0)
        if kind == "timed-tape+1sync":
            eng.step_tape_device_timed(base, TAPE, t_first % TAPE, K, 0, 1)
        else:
            eng.step_tape_device(base, TAPE, t_first % TAPE, K)
        if kind in ("events+2sync", "events+1sync"):
            eng.event_record(1)
        t1 = time.perf_counter()
        if kind == "events+2sync":
            torch.cuda.synchronize()
            torch.cuda.synchronize()
        elif kind in ("events+1sync", "torch-sync", "timed-tape+1sync"):
            torch.cuda.synchronize()
        elif kind == "hipDeviceSynchronize":
            hip.hipDeviceSynchronize()
        elif kind == "rw_sync":
            eng.sync()
        t2 = time.perf_counter()
        ev = eng.event_elapsed_ms(0, 1) * 1e3 if kind.startswith(("events", "timed")) else float("nan")
        return (t1 - t0) * 1e6, (t2 - t0) * 1e6, ev

    t = 200
    print(f"K={K} launches, B={B}, {TRIALS} regions each; HSA_ENABLE_INTERRUPT={os.environ.get('HSA_ENABLE_INTERRUPT')}")
    print(f"{'bracket':24s} {'submit us':>10s} {'total us':>10s} {'min total':>10s} {'us/step':>8s} {'events us':>10s}")
    for kind in ("events+2sync", "events+1sync", "timed-tape+1sync", "torch-sync", "hipDeviceSynchronize", "rw_sync", "events+2sync", "timed-tape+1sync"):
        sub, tot, evs = [], [], []
        for _ in range(TRIALS):
            a, b, c = region(kind, t)
            t += K
            sub.append(a)
            tot.append(b)
            evs.append(c)
        print(f"{kind:24s} {statistics.median(sub):10.1f} {statistics.median(tot):10.1f} {min(tot):10.1f} {statistics.median(tot) / K:8.3f} {statistics.median(evs):10.1f}")
    # the first region after start-up, as the driver's `--steps 20 --warmup 5` sees it, is in bench.py's own line


if __name__ == "__main__":
    main()
