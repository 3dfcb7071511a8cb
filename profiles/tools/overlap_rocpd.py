#!/usr/bin/env python
"""Kernel-trace view of the two-pipeline run: per queue the step kernel's dispatch count and average duration, and how much of
the busy time two step kernels (of different queues) were in flight together.
    rocprofv3 --kernel-trace -d gpurun_out/pipes -o t -- python profiles/tools/split_probe.py rware-small-10ag-v1 16384 2
    python profiles/tools/overlap_rocpd.py gpurun_out/pipes > profiles/rNN_two_pipelines_trace.txt"""
import glob
import os
import sqlite3
import sys

root = sys.argv[1] if len(sys.argv) > 1 else "."
for db in sorted(glob.glob(os.path.join(root, "**", "*_results.db"), recursive=True)):
    con = sqlite3.connect(db)
    rows = con.execute("select queue, stream, start, end, grid_x from kernels where name like '%rware_step_kernel%' order by start").fetchall()
    con.close()
    if not rows:
        continue
    print(f"== {db}")
    per = {}
    for qn, st, a, b, g in rows:
        per.setdefault((qn, st, g), []).append((a, b))
    for (qn, st, g), v in sorted(per.items()):
        d = sorted(b - a for a, b in v)
        print(f"  {qn:10s} {st:12s} grid {g:8d} dispatches {len(v):6d}  avg {sum(d) / len(d) / 1e3:8.3f} us  median {d[len(d) // 2] / 1e3:8.3f} us")
    # the half-batch launches (the two pipelines), the last 60 % of them (the warm-up calls in front run one engine at a time)
    gmin = min(r[4] for r in rows)
    rows = [r for r in rows if r[4] == gmin]
    rows = rows[int(len(rows) * 0.4):]
    ev = sorted([(a, 1) for _, _, a, b, _ in rows] + [(b, -1) for _, _, a, b, _ in rows])
    depth, last, busy, both = 0, ev[0][0], 0, 0
    for t, d in ev:
        if depth >= 1:
            busy += t - last
        if depth >= 2:
            both += t - last
        depth += d
        last = t
    span = ev[-1][0] - ev[0][0]
    print(f"  two pipelines, steady part: {len(rows)} half-batch dispatches over {span / 1e3:.1f} us; some step kernel in flight {busy / span:.1%} of it, two or more "
          f"{both / span:.1%}; wall per dispatch pair {2 * span / len(rows) / 1e3:.3f} us")
