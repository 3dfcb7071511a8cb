"""TEST / MEASUREMENT INFRASTRUCTURE — the CPU baselines bench.py reports beside the GPU number.

Two legs, both bounded in time, both on the same workload as the headline (env id, uniform random
actions, step + FLATTENED observation, reset when an episode ends):

  reference  the UNMODIFIED `rware.warehouse.Warehouse.step` (`/root/reference/rware/warehouse.py:804-946`), one
             env per process, 1 process and P processes (multiprocessing, one per host core) — what
             BASELINE.json's `north_star` and SURVEY.md §8(d) name as the baseline.  Imported from /root/reference
             in the build container, and from the byte-for-byte staged copy under the git-ignored oracle/_ref/
             (oracle/make_ref.sh, run by __graft_entry__.build()) on the GPU box, which has no reference tree.
  port       `oracle/rware_oracle.c` (the C restatement of the same step), 1 process and P processes, each
             single-threaded with its own batch of envs.

Nothing here is product code; only bench.py's `cpu_baseline` leg and profiles/tools call it.
"""
from __future__ import annotations

import multiprocessing as mp
import os
import sys
import time

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
for _p in (ROOT, _HERE):
    if _p not in sys.path:
        sys.path.insert(0, _p)


def cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform

    return platform.processor() or platform.machine() or "unknown"


def host_cores() -> int:
    try:
        return max(1, len(os.sched_getaffinity(0)))
    except AttributeError:
        return max(1, os.cpu_count() or 1)


def cpu_quota():
    """CPUs' worth of time the container may use per period (cgroup v2 `cpu.max` / v1 `cpu.cfs_quota_us`), or None when
    unlimited / unreadable.  The GPU box reports 256 schedulable CPUs, yet 256 single-threaded processes ran only ~16x as
    fast as one (round 3: both the C port and the Python reference) — a quota is the usual reason; it is reported so the
    aggregate is not read as a 256-core figure."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            q = float(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            per = float(f.read())
        return None if q <= 0 else q / per
    except (OSError, ValueError):
        return None


# ------------------------------------------------------------------------------------------ process fan-out
def _fan_out(worker, args, procs, seconds):
    """Runs `worker(*args, seed, queue, (ready_semaphore, go_event))` in `procs` forked processes; each reports (agent_steps, seconds)."""
    ctx = mp.get_context("fork")
    q, go, ready = ctx.Queue(), ctx.Event(), ctx.Semaphore(0)
    ps = [ctx.Process(target=worker, args=args + (1000 * i, q, (ready, go))) for i in range(procs)]
    for p in ps:
        p.start()
    for _ in ps:  # every process has built its envs and warmed up (or failed) before the clock starts
        ready.acquire(timeout=120)
    go.set()
    got = [q.get(timeout=seconds * 4 + 120) for _ in ps]
    for p in ps:
        p.join(30)
    for g in got:
        if isinstance(g, Exception):
            raise g
    return sum(g[0] for g in got) / max(g[1] for g in got)


# ------------------------------------------------------------------------------------------ C port
def _port_worker(env_id, b, seconds, seed, q, go):
    try:
        import rware_amd
        from rware_oracle import OracleVecEnv

        kw = rware_amd.env_kwargs(env_id)
        kw["reward_type"] = kw["reward_type"].value
        env = OracleVecEnv(b, **kw)
        env.reset(seed=seed)
        acts = np.random.default_rng(12345 + seed).integers(0, 5, size=(64, b, kw["n_agents"]), dtype=np.int32)
        for t in range(4):
            env.step_autoreset(acts[t], "next_step")
    except Exception as exc:  # report instead of leaving the parent to time out
        q.put(exc)
        go[0].release()
        return
    go[0].release()
    go[1].wait()
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for t in range(8):
            env.step_autoreset(acts[(n + t) % 64], "next_step")
        n += 8
    q.put((b * kw["n_agents"] * n, time.perf_counter() - t0))


def time_port(env_id: str, seconds: float, procs: int, b: int = 512):
    """agent-steps/s of the C oracle on `procs` host processes (each with its own `b` envs, single-threaded)."""
    from rware_oracle import lib

    lib()  # build once, before forking
    return _fan_out(_port_worker, (env_id, b, seconds), procs, seconds)


# ------------------------------------------------------------------------------------------ the reference itself
def _ref_worker(env_id, seconds, seed, q, go):
    import ref_runner as rr

    try:
        env = rr.make_reference_env(env_id.replace("-v1", "-v2"))  # this snapshot registers the -v2 ids (rware/__init__.py:22-39)
        env.reset(seed=seed)
        n_agents = env.n_agents
        acts = np.random.default_rng(12345 + seed).integers(0, 5, size=(4096, n_agents)).tolist()
        for t in range(20):
            env.step(acts[t])
    except Exception as exc:
        q.put(exc)
        go[0].release()
        return
    go[0].release()
    go[1].wait()
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for t in range(50):
            _, _, done, _, _ = env.step(acts[(n + t) % 4096])   # the unpatched reference step, stock networkx
            if done:
                env.reset()
        n += 50
    q.put((n_agents * n, time.perf_counter() - t0))


def reference_available() -> bool:
    import ref_runner as rr

    return rr.reference_available()


def time_reference(env_id: str, seconds: float, procs: int):
    """agent-steps/s of the unmodified reference, `procs` processes with one env each."""
    import ref_runner as rr

    rr.load_reference()
    return _fan_out(_ref_worker, (env_id, seconds), procs, seconds), rr.using_standin_gymnasium()


def measure(env_id: str, port_seconds: float = 5.0, ref_seconds: float = 6.0) -> dict:
    """The `cpu_baseline` object of bench.py's JSON line."""
    P_sched, model, quota = host_cores(), cpu_model(), cpu_quota()
    # processes of the aggregate legs: one per CPU the job can actually run on at once — the schedulable set, capped by
    # the container's CPU quota (more single-threaded processes than that only time-slice)
    P = P_sched if not quota else max(1, min(P_sched, int(quota + 0.5)))
    port_1 = time_port(env_id, port_seconds, 1)
    port_p = time_port(env_id, port_seconds, P) if P > 1 else port_1
    out = {
        "unit": "agent-steps/s", "cores": P, "cpu_model": model, "cpu_quota_cores": quota, "schedulable_cpus": P_sched,
        "port": {"single": port_1, "aggregate": port_p, "processes": P,
                 "what": "oracle/rware_oracle.c (C restatement of Warehouse.step + FLATTENED obs), 512 envs per process"},
    }
    if reference_available():
        ref_1, standin = time_reference(env_id, ref_seconds, 1)
        ref_p, _ = time_reference(env_id, ref_seconds, P) if P > 1 else (ref_1, standin)
        import ref_runner as rr

        out.update(kind="reference", value=ref_p, single=ref_1, aggregate=ref_p, processes=P, parallel_speedup=ref_p / ref_1,
                   reference_root=("oracle/_ref (staged by oracle/make_ref.sh)" if rr.REFERENCE_ROOT == rr.STAGED_ROOT else rr.REFERENCE_ROOT),
                   sample=f"{env_id}: unmodified rware.warehouse.Warehouse.step (pure Python + networkx), one env per process, "
                          f"uniform random actions, reset on done, ~{ref_seconds:.0f} s with 1 process and ~{ref_seconds:.0f} s with "
                          f"{P} processes on {P} cores of '{model}'"
                          + (" (gymnasium stand-in: the real package is not installed)" if standin else ""))
    else:
        out.update(kind="port", value=port_p, single=port_1, aggregate=port_p,
                   sample=f"{env_id}: C port of the reference step, uniform random actions, next_step autoreset, ~{port_seconds:.0f} s with "
                          f"1 process and ~{port_seconds:.0f} s with {P} processes on {P} cores of '{model}' (the Python reference "
                          "is not present on this box; its committed measurement is under `reference_python_recorded`)")
        rec = os.path.join(ROOT, "profiles", "cpu_reference_python.json")
        if os.path.exists(rec):
            import json

            try:
                out["reference_python_recorded"] = json.load(open(rec))
            except Exception:
                pass
    return out


if __name__ == "__main__":
    import json

    env_id = sys.argv[1] if len(sys.argv) > 1 else "rware-small-4ag-v1"
    print(json.dumps(measure(env_id), indent=1))
