#!/usr/bin/env python
"""Reads the rocprofv3 databases profiles/tools/sweep.sh produced and writes
  gpurun_out/<tag>_sweep.json   one record per config: the bench.py JSON line of the trace run, the kernel-trace average
                                duration of the step kernel, FETCH_SIZE / WRITE_SIZE per launch, the roofline fractions on
                                algorithmic and on physical bytes (vs 8 TB/s and vs the 6.29 TB/s copy ceiling)
  gpurun_out/<tag>_sweep.txt    the same as a table + the per-database summaries (what gets committed under profiles/)
  gpurun_out/pmc_traffic.json   physical bytes per launch keyed the way bench.py looks them up, stamped with the hash of
                                the kernel sources they were measured on
Usage: sweep_collect.py <sweep dir> <tag>"""
import glob
import json
import os
import sqlite3
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def rows(db, sql):
    con = sqlite3.connect(db)
    try:
        return con.execute(sql).fetchall()
    except sqlite3.Error:
        return []
    finally:
        con.close()


def find_db(d):
    got = sorted(glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True))
    return got[0] if got else None


def step_kernel_trace(db):
    """(name, calls, avg us) of the step kernel with the most calls."""
    best = None
    for name, calls, total, avg, pct in rows(db, "select name, total_calls, total_duration, average, percentage from top_kernels"):
        if "rware_step_kernel" in name and (best is None or calls > best[1]):
            best = (name, calls, avg)
    return best


def step_kernel_dispatches(db):
    """Per-dispatch view of the most-called step kernel: rocprofv3 stamps a dispatch that starts right behind its
    predecessor with start == the predecessor's end, so its duration is the whole launch-to-launch interval; a dispatch
    the host (slowed by the profiler) submitted late starts after an idle gap on a cold chip and runs longer.  Returns the
    statistics of both groups."""
    per = {}
    for name, start, end in rows(db, "select name, start, end from kernels order by start"):
        if "rware_step_kernel" in name:
            per.setdefault(name, []).append((start, end))
    if not per:
        return None
    name = max(per, key=lambda k: len(per[k]))
    se = per[name][20:]  # (skip the reset launch and the first warm-up launches)
    if len(se) < 10:
        return None
    dur = [(e - s) / 1000.0 for s, e in se]
    gap = [0.0] + [(se[i + 1][0] - se[i][1]) / 1000.0 for i in range(len(se) - 1)]
    b2b = [d for d, g in zip(dur, gap) if g < 0.2]
    late = [d for d, g in zip(dur, gap) if g >= 0.2]
    return {"dispatches": len(dur), "mean_us": statistics.mean(dur), "median_us": statistics.median(dur),
            "back_to_back": {"n": len(b2b), "median_us": statistics.median(b2b) if b2b else None,
                             "mean_us": statistics.mean(b2b) if b2b else None},
            "after_an_idle_gap": {"n": len(late), "median_us": statistics.median(late) if late else None,
                                  "median_gap_us": statistics.median([g for g in gap if g >= 0.2]) if late else None}}


def step_kernel_pmc(db, counter):
    """median per-dispatch value (KiB) of `counter` over the dispatches of the most-dispatched step kernel, + its resources"""
    per = {}
    for name, val, vg, sg, lds, wg, grid in rows(db, "select kernel_name, value, vgpr_count, sgpr_count, lds_block_size, workgroup_size, "
                                                     f"grid_size from counters_collection where counter_name = '{counter}'"):
        if "rware_step_kernel" in name or "copy" in name:
            per.setdefault(name, []).append((val, vg, sg, lds, wg, grid))
    if not per:
        return None
    step = [k for k in per if "rware_step_kernel" in k]
    name = max(step or list(per), key=lambda k: len(per[k]))
    vals = [v[0] for v in per[name]]
    r = per[name][0]
    return {"kernel": name, "dispatches": len(vals), "median_KiB": statistics.median(vals), "min_KiB": min(vals), "max_KiB": max(vals),
            "vgpr": r[1], "sgpr": r[2], "lds": r[3], "workgroup": r[4], "grid": r[5]}


def calib(out):
    res = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        db = find_db(os.path.join(out, "calib", f"pmc_{c}"))
        if not db:
            continue
        for name, val in rows(db, f"select kernel_name, value from counters_collection where counter_name = '{c}'"):
            key = "copy16" if "copy16" in name else "copy4" if "copy4" in name else None
            if key:
                res.setdefault(f"{key}_{c}_KiB", []).append(val)
    return {k: statistics.median(v) for k, v in res.items()}


def main(out, tag):
    import bench

    sha = bench.kernel_sources_sha()
    cal = calib(out)
    gib = 1024.0 * 1024.0  # the calibration kernels move 1 GiB each way = 1048576 KiB
    factors = {}
    for key in ("copy16", "copy4"):
        if f"{key}_FETCH_SIZE_KiB" in cal:
            factors[f"{key}_fetch_reported_over_true"] = cal[f"{key}_FETCH_SIZE_KiB"] / gib
        if f"{key}_WRITE_SIZE_KiB" in cal:
            factors[f"{key}_write_reported_over_true"] = cal[f"{key}_WRITE_SIZE_KiB"] / gib
    records, traffic = [], {}
    for d in sorted(p for p in glob.glob(os.path.join(out, "*")) if os.path.isdir(p) and os.path.basename(p) != "calib"):
        name = os.path.basename(d)
        rec = {"config": name}
        try:
            line = [l for l in open(os.path.join(d, "trace.out")).read().splitlines() if l.startswith("{")][-1]
            b = json.loads(line)
        except Exception as exc:  # noqa: BLE001
            rec["error"] = f"no bench line: {exc}"
            records.append(rec)
            continue
        rec["bench"] = {k: b[k] for k in ("value", "ms_per_step", "steps", "warmup")}
        rec["bench"]["workload"] = b["config"]["workload"]
        rec["bench"]["kernel_specialised"] = b["config"]["kernel_specialised"]
        rec["bench"]["submit"] = b["config"]["submit"]
        rec["bench"]["envs_per_workgroup"] = b["config"]["envs_per_workgroup"]
        rec["event_ms_per_step"] = b["roofline"]["kernel_ms_per_launch"]
        a_bytes = b["roofline"]["algorithmic_bytes_per_launch"]
        rec["algorithmic_bytes_per_launch"] = a_bytes
        e_bytes = b["roofline"].get("engine_bytes_per_launch")  # what the engine's layout has to move (rw_info)
        rec["engine_bytes_per_launch"] = e_bytes
        db = find_db(os.path.join(d, "trace"))
        kt = step_kernel_trace(db) if db else None
        steps_per_launch = 64 if "fused" in name else 1
        if kt:
            rec["kernel_trace"] = {"kernel": kt[0][:100], "calls": kt[1], "avg_us": kt[2], "us_per_step": kt[2] / steps_per_launch}
            disp = step_kernel_dispatches(db)
            if disp:
                rec["kernel_trace"]["per_dispatch"] = disp
                if disp["back_to_back"]["median_us"]:
                    rec["kernel_trace"]["us_per_step_back_to_back"] = disp["back_to_back"]["median_us"] / steps_per_launch
        pm = {}
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            db = find_db(os.path.join(d, f"pmc_{c}"))
            if db:
                pm[c] = step_kernel_pmc(db, c)
        rec["pmc"] = pm
        us = rec.get("kernel_trace", {}).get("us_per_step")
        us_b2b = rec.get("kernel_trace", {}).get("us_per_step_back_to_back")
        if us_b2b:
            rec["roofline_on_algorithmic_bytes_back_to_back"] = {"GBps": a_bytes / us_b2b / 1e3, "frac_of_8TBps": a_bytes / us_b2b / 1e3 / 8000.0}
        if us:
            rec["roofline_on_algorithmic_bytes"] = {"GBps": a_bytes / us / 1e3, "frac_of_8TBps": a_bytes / us / 1e3 / 8000.0,
                                                    "frac_of_6.29TBps": a_bytes / us / 1e3 / 6290.0}
        if pm.get("FETCH_SIZE") and pm.get("WRITE_SIZE"):
            f, w = pm["FETCH_SIZE"]["median_KiB"], pm["WRITE_SIZE"]["median_KiB"]
            # MI355X_MICROARCH.md §HBM: FETCH_SIZE reports 1/2 of a wide coalesced read -> doubled; WRITE_SIZE is exact
            # (both re-checked by the calibration run above).  Per launch; a fused launch covers 64 steps.
            phys = (2.0 * f + w) * 1024.0 / steps_per_launch
            rec["physical_bytes_per_step"] = {"corrected": phys, "lower_bound_uncorrected_fetch": (f + w) * 1024.0 / steps_per_launch,
                                              "fetch_KiB_reported": f, "write_KiB": w}
            if e_bytes and steps_per_launch == 1:
                # the self-check of the roofline line: the PMC traffic of a step must be what the layout says it moves
                ratio = phys / e_bytes
                rec["physical_over_engine_bytes"] = ratio
                rec["physical_matches_engine_bytes"] = bool(0.95 <= ratio <= 1.10)
            if us:
                rec["roofline_on_engine_bytes"] = {"GBps": e_bytes / us / 1e3, "frac_of_8TBps": e_bytes / us / 1e3 / 8000.0} if e_bytes else None
                rec["roofline_on_physical_bytes"] = {"GBps": phys / us / 1e3, "frac_of_8TBps": phys / us / 1e3 / 8000.0,
                                                     "frac_of_6.29TBps": phys / us / 1e3 / 6290.0}
            if steps_per_launch == 1:
                wl = b["config"]["workload"].split()[0]
                batch = b["config"]["envs_per_gpu"]
                key = f"{wl}:{batch}"
                if "r2" in name:
                    key += ":r2"
                if "image" in name:
                    key += ":obs2"
                if "msg2" in name:
                    key += ":m2"
                traffic[key] = {"bytes_per_launch": int(phys), "fetch_KiB_reported": f, "write_KiB": w, "config": name}
        records.append(rec)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", f"{tag}_sweep.json"), "w") as fh:
        json.dump({"tag": tag, "kernel_sources_sha": sha, "calibration": {"raw": cal, **factors}, "configs": records}, fh, indent=1)
    with open(os.path.join(ROOT, "gpurun_out", "pmc_traffic.json"), "w") as fh:
        json.dump({"kernel_sources_sha": sha, "measured_by": f"profiles/tools/sweep.sh {tag}", "calibration": factors,
                   "_note": "physical bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) KiB * 1024, medians over the dispatches of separate "
                            "rocprofv3 --pmc passes; the factor 2 on FETCH_SIZE is the gfx950 under-count of wide reads "
                            "(MI355X_MICROARCH.md, re-measured by copy_calib.hip in the same sweep)",
                   "entries": traffic}, fh, indent=1)
    lines = [f"sweep {tag}: kernel sources {sha}", f"calibration: {json.dumps({**cal, **factors})}", "",
             "us/step: 'trace avg' = rocprofv3 --stats average; 'b2b' = median of the dispatches that started right behind their predecessor",
             "(the others were submitted late by the profiled host and ran on an idle, cold chip); 'events' = HIP events in the same profiled run.",
             f"{'config':28s} {'trace avg':>10s} {'b2b':>8s} {'late n':>7s} {'events':>8s} {'A MB':>8s} {'engine MB':>9s} {'phys MB':>8s} {'phys/eng':>8s} {'frac A/8T':>9s} {'frac eng/8T':>11s} {'frac phys/8T':>12s} {'phys/6.29T':>10s}"]
    for r in records:
        if "error" in r:
            lines.append(f"{r['config']:28s} {r['error']}")
            continue
        us = r.get("kernel_trace", {}).get("us_per_step", float("nan"))
        ph = r.get("physical_bytes_per_step", {}).get("corrected", float("nan"))
        fa = r.get("roofline_on_algorithmic_bytes", {}).get("frac_of_8TBps", float("nan"))
        fp = r.get("roofline_on_physical_bytes", {}).get("frac_of_8TBps", float("nan"))
        fm = r.get("roofline_on_physical_bytes", {}).get("frac_of_6.29TBps", float("nan"))
        kt = r.get("kernel_trace", {})
        b2b = kt.get("us_per_step_back_to_back", float("nan"))
        late = kt.get("per_dispatch", {}).get("after_an_idle_gap", {}).get("n", 0)
        eb = r.get("engine_bytes_per_launch") or float("nan")
        fe = (r.get("roofline_on_engine_bytes") or {}).get("frac_of_8TBps", float("nan"))
        ratio = r.get("physical_over_engine_bytes", float("nan"))
        flag = "" if r.get("physical_matches_engine_bytes", True) else "  <-- outside [0.95, 1.10]"
        lines.append(f"{r['config']:28s} {us:10.3f} {b2b:8.3f} {late:7d} {r['event_ms_per_step'] * 1e3:8.3f} {r['algorithmic_bytes_per_launch'] / 1e6:8.2f} "
                     f"{eb / 1e6:9.2f} {ph / 1e6:8.2f} {ratio:8.3f} {fa:9.3f} {fe:11.3f} {fp:12.3f} {fm:10.3f}{flag}")
    bad = [r["config"] for r in records if r.get("physical_matches_engine_bytes") is False]
    lines.append("")
    lines.append("physical (PMC) bytes / engine bytes within [0.95, 1.10] for every per-step config" if not bad
                 else f"PHYSICAL TRAFFIC DOES NOT MATCH THE ENGINE'S BYTE MODEL: {', '.join(bad)}")
    with open(os.path.join(ROOT, "gpurun_out", f"{tag}_sweep.txt"), "w") as fh:
        fh.write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "sweep")
