#!/usr/bin/env python
"""Stage timeline of the chunk-pipelined persistent step kernel (rw_debug_timeline with a PIPE build): per chunk `it` < 4 of every
workgroup, the stamps around the two barriers and the end of every wavefront's share.  Usage: pipe_timeline.py ENV_ID B [sensor_range].
RWARE_PIPE_E / RWARE_PIPE_WGS_PER_CU pick the geometry."""
import os
import sys

import numpy as np

os.environ.setdefault("RWARE_HOOKS", "1")  # (this tool drives the library's A/B hooks: csrc/rware_hooks.h)

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import rware_amd  # noqa: E402

TL_PIPE = 24   # (rware_kernels.h; slot 17 = TL_PIPE_FIRST_AG: the agent phases of chunk 0, which run in the prologue)


def main():
    env_id, B = sys.argv[1], int(sys.argv[2])
    kw = rware_amd.env_kwargs(env_id)
    if len(sys.argv) > 3:
        kw["sensor_range"] = int(sys.argv[3])
    env = rware_amd.WarehouseVecEnv(B, pipe=True, **kw)
    eng = env.engines[0]
    i = eng.info
    assert i.pipe_workgroups, "no pipelined build for this shape"
    env.reset(seed=0)
    acts = torch.randint(0, 5, (32, B, kw["n_agents"]), dtype=torch.int32).cuda()
    for t in range(20):
        eng.step_device(acts[t].data_ptr())
    eng.sync()
    raw = eng.debug_timeline(acts[21].data_ptr()).astype(np.int64)
    t0 = raw[:, 0].min()
    us = lambda c: (raw[:, c] - t0) / 100.0  # noqa: E731
    n_chunks = B // i.pipe_envs_per_workgroup
    print(f"{env_id} B={B} pipe E={i.pipe_envs_per_workgroup} wgs={i.pipe_workgroups} ({i.pipe_workgroups / i.compute_units:.1f} per CU), "
          f"{n_chunks / i.pipe_workgroups:.2f} chunks per workgroup, lds {2 * 0} ")
    med = lambda a: f"{np.median(a):6.2f} (p90 {np.percentile(a, 90):6.2f})"  # noqa: E731
    print(f"start {med(us(0))}   staged (prologue) {med(us(4))}   agent phases of chunk 0 done {med(us(17))}   end {med(us(9))}  kernel span {us(9).max():.2f}")
    for it in range(4):
        b = TL_PIPE + 8 * it
        if raw[:, b].max() == 0:
            break
        ok = raw[:, b] > 0
        A, g0, w3, Bb, ag, ex, dm = (us(b + k)[ok] for k in range(7))
        nxt = raw[:, b + 8] > 0 if b + 8 < raw.shape[1] else np.zeros_like(ok)
        print(f"chunk {it}: {ok.sum()} workgroups | behind A at {med(A)} | A->gather done {med(g0 - A)} | A->self bits + write-back done {med(w3 - A)} | "
              f"A->B {med(Bb - A)}")
        has_ag = raw[:, b + 4][ok] > 0
        print(f"         B->agent phases of chunk {it + 1} done {med((ag - Bb)[has_ag]) if has_ag.any() else '   -'} | B->expansion issued {med(ex - Bb)} | "
              f"B->stage-in of chunk {it + 2} issued {med(dm - Bb)}")
        both = ok & nxt
        if both.any():
            A2 = us(b + 8)[both]
            landed = us(b + 7)[both]
            print(f"         B->next A {med(A2 - us(b + 3)[both])} | stage-in landed {med(landed - A2)} behind the next A  => stage period {med(A2 - us(b)[both])}")
    env.close()


if __name__ == "__main__":
    main()
