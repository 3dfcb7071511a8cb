#!/usr/bin/env python
"""Headline benchmark: agent-steps/s of the HIP step engine on rware-small-4ag, batch 16384 per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W]          # N > 1: spawns one process per GPU itself
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   # or under a launcher

One "step" = one pass of the hot path (Warehouse.step + FLATTENED obs, rware/warehouse.py:804-946)
over the whole env batch resident in HBM: one `rw_step_device` call, actions read from a
device-resident tape, NEXT_STEP autoreset on (all envs reset on-device every 500 steps).
Multi-GPU: one process per GPU, each with its own 16384-env shard (weak scaling), seeds offset by
the global env index, NO data-path collective and NO RCCL (envs are independent);
the only communication is a barrier and a MAX over ranks of the timings, on CPU tensors over gloo.

The printed JSON line carries
  value / ms_per_step   exactly K steps after W warm-up steps, barrier + device sync on both sides
  sustained             the same launches for a fixed 2000 steps after 100 warm-up steps (SURVEY.md §8(d) protocol,
                        spans 4 mass resets) — the steady state, whatever K and W the caller chose
  soak                  the same launches back to back for >= 3 s (steps, ms_per_step, wall_s): visible to a utilisation sampler outside
                        this process, checkable against the caller's own clock
  roofline              bytes per launch / HIP-event time per launch on the engine's stream (the two events ride on the first
                        and the last launch of the timed region: rw_step_tape_device_timed) vs the 8 TB/s HBM peak.
                        `achieved` / `frac` are the PHYSICAL bandwidth: `traffic` (bytes per launch from the rocprofv3 PMC
                        passes, profiles/pmc_traffic.json, tied to the kernel sources by hash) when that record is current,
                        else the bytes this engine's layout has to move (`basis` says which) — never above 1.
                        `achieved_algorithmic` / `frac_algorithmic` price SURVEY.md §8(d)'s algorithmic bytes, which the engine
                        does not move (a work rate in that unit; it may exceed 1).  `regime`: whether a step's bytes fit the
                        256 MiB Infinity Cache ("infinity-cache") or not ("hbm")
  (--submit graph: the same per-step launches replayed from a HIP graph — for rocprofv3 traces, see profiles/tools/sweep.sh)
  cpu_baseline (N = 1)  the reference's pure-Python step on the host cores when /root/reference exists, else the C port
                        of it, 1 process and one per core, with the core count and CPU model
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ENV_ID = "rware-small-4ag-v1"
BATCH_PER_GPU = 16384
HBM_PEAK_GBPS = 8000.0       # MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_MEASURED_GBPS = 6290.0   # same guide: measured float4 copy ceiling
INFINITY_CACHE_BYTES = 256 * 1024 * 1024  # same guide: 256 MiB memory-side cache in front of HBM


def roofline_fields(a_bytes, e_bytes, traffic, k_ms):
    """The bandwidth fields of one measured kernel time.  `achieved` / `frac`: the physical figure — PMC traffic when the recorded pass
    belongs to these kernel sources, else the engine-layout bytes (`basis`); `*_algorithmic`: SURVEY.md §8(d)'s bytes (a work rate);
    `regime`: do a step's bytes fit the Infinity Cache.  No key named frac* other than frac_algorithmic can exceed 1."""
    sec = k_ms * 1e-3
    phys = traffic if traffic else e_bytes
    return {
        "achieved": phys / sec / 1e9, "frac": phys / sec / 1e9 / HBM_PEAK_GBPS, "basis": "pmc-traffic" if traffic else "engine-bytes",
        "regime": "hbm" if phys > INFINITY_CACHE_BYTES else "infinity-cache",
        "achieved_algorithmic": a_bytes / sec / 1e9, "frac_algorithmic": a_bytes / sec / 1e9 / HBM_PEAK_GBPS,
        "achieved_engine": e_bytes / sec / 1e9, "frac_engine": e_bytes / sec / 1e9 / HBM_PEAK_GBPS,
        "traffic": traffic, "traffic_over_engine_bytes": (traffic / e_bytes) if traffic else None,
        "frac_physical": (traffic / sec / 1e9 / HBM_PEAK_GBPS) if traffic else None,
        "algorithmic_bytes_per_launch": a_bytes, "engine_bytes_per_launch": e_bytes,
        "peak": HBM_PEAK_GBPS, "peak_measured": HBM_MEASURED_GBPS,
    }
def floor_fraction(floor_ms, k_ms):
    """store-only kernel time / step kernel time.  A kernel that only writes the observations cannot take longer than the one that also
    computes them: a ratio above 1 means the two measurements were disturbed differently (ranks sharing one device in the rehearsal tests, a
    clock step between the two legs) — reported as None rather than as a fraction above 1."""
    if not floor_ms or not k_ms:
        return None
    f = floor_ms / k_ms
    return f if f <= 1.0 else None


TAPE_STEPS = int(os.environ.get("RWARE_BENCH_TAPE_STEPS", "256"))
SUSTAINED_STEPS, SUSTAINED_WARMUP = 2000, 100
SOAK_SECONDS = 3.2  # the `soak` leg: back-to-back launches for at least this long (an outside utilisation sampler can see the GPU busy)
try:
    ORIG_AFFINITY = set(os.sched_getaffinity(0))  # before pin_rank() narrows it
except AttributeError:
    ORIG_AFFINITY = None
# everything that decides what a step moves: the kernels AND the host side's choices (store mode, build / geometry selection)
KERNEL_SOURCES = ("rware_kernels.h", "rware_phase_stage_in.h", "rware_phase_pipe.h", "rware_phase_goals.h", "rware_phase_agents_reg.h",
                  "rware_phase_agents_lds.h", "rware_phase_reset.h", "rware_phase_write_back.h", "rware_phase_gather.h", "rware_phase_expand.h", "rware_cdna4.h", "rware_pcg64.h", "rware_static_table.h", "rware_static.hip", "rware_generic.hip",
                  "rware_kernel_table.h", "rware_capi.hip")


def strip_cxx_comments(text: str) -> str:
    """C++ source without // and /* */ comments (string and character literals respected), runs of white space collapsed, blank lines
    dropped: what the compiler sees.  A comment or layout edit leaves it unchanged."""
    out, i, n = [], 0, len(text)
    while i < n:
        c = text[i]
        if c in "\"'":  # a literal: copy it through verbatim
            j = i + 1
            while j < n and text[j] != c:
                j += 2 if text[j] == "\\" else 1
            out.append(text[i:j + 1])
            i = j + 1
        elif text.startswith("//", i):
            while i < n and text[i] != "\n":
                i += 2 if (text[i] == "\\" and i + 1 < n and text[i + 1] == "\n") else 1  # (a line comment continues behind a backslash-newline)
        elif text.startswith("/*", i):
            j = text.find("*/", i + 2)
            i = n if j < 0 else j + 2
            out.append(" ")
        else:
            out.append(c)
            i += 1
    lines = (" ".join(ln.split()) for ln in "".join(out).splitlines())
    return "\n".join(ln for ln in lines if ln)


def kernel_sources_sha() -> str:
    """Identifies the kernels a PMC traffic figure was measured on (profiles/pmc_traffic.json carries the same hash; a CPU test,
    tests/test_host_layer.py, fails while the two differ).  Hashes the CODE of the kernel sources — comments and white space are
    stripped first, so a comment edit behind the evidence pass does not orphan the evidence (round 5 lost `roofline.traffic` that way)."""
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "robotic-warehouse_amd", "csrc", name), "r", encoding="utf-8") as f:
            h.update(name.encode() + b"\0" + strip_cxx_comments(f.read()).encode() + b"\0")
    return h.hexdigest()[:16]


def cpu_baseline(env_id: str):
    """CPU legs in a fresh interpreter (it forks workers; this process holds an initialised HIP runtime)."""
    cmd = [sys.executable, os.path.join(ROOT, "oracle", "cpu_baseline.py"), env_id]
    # (this process pinned itself to a few cores next to its GPU: the CPU legs get every core the job was given)
    unpin = (lambda: os.sched_setaffinity(0, ORIG_AFFINITY)) if ORIG_AFFINITY else None
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, preexec_fn=unpin)
    if out.returncode != 0:
        return {"error": out.stderr[-500:]}
    return json.loads(out.stdout)


def spawn_ranks(n: int):
    """`python bench.py --gpus N` without a launcher: this process becomes rank 0 and starts ranks 1..N-1."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    base = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n))
    children = []
    for r in range(1, n):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        children.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=subprocess.DEVNULL))
    os.environ.update(base, RANK="0", LOCAL_RANK="0")
    import atexit

    # if this rank dies early, the others would sit at the rendezvous until its timeout: take them down with it
    atexit.register(lambda: [c.kill() for c in children if c.poll() is None])
    return children


# ------------------------------------------------------------------------------------------ host placement
def _parse_cpulist(txt):
    out = []
    for part in txt.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out.extend(range(int(a), int(b or a) + 1))
    return out


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def gpu_numa_node(torch, dev):
    """NUMA node of HIP device `dev` from its PCI address (/sys/bus/pci/devices/<bdf>/numa_node), or -1."""
    try:
        p = torch.cuda.get_device_properties(dev)
        bdf = f"{int(p.pci_domain_id):04x}:{int(p.pci_bus_id):02x}:{int(p.pci_device_id):02x}.0"
        v = _read(f"/sys/bus/pci/devices/{bdf}/numa_node")
        return int(v) if v is not None else -1
    except Exception:  # noqa: BLE001
        return -1


def pin_rank(torch, local_rank, local_world, dev_of_rank):
    """Pins this process to whole physical cores of its GPU's NUMA node, disjoint from the other local ranks' cores.
    The step kernel is ~7 us and a launch costs the host ~3.6 us: a rank that shares SMT siblings with another rank, or sits
    on the far socket from its GPU, is how 8 x 9 G agent-steps/s becomes 5 x.  Returns a description for the JSON line."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except AttributeError:
        return {"pinned": False, "why": "no sched_getaffinity"}
    if os.environ.get("RWARE_BENCH_NO_PIN") == "1":
        return {"pinned": False, "why": "RWARE_BENCH_NO_PIN=1", "cpus": len(allowed)}
    nodes = [gpu_numa_node(torch, dev_of_rank(r)) for r in range(local_world)]
    mine = nodes[local_rank]
    cpus = allowed
    if mine >= 0:
        txt = _read(f"/sys/devices/system/node/node{mine}/cpulist")
        on_node = [c for c in (_parse_cpulist(txt) if txt else []) if c in set(allowed)]
        if on_node:
            cpus = on_node
        else:
            mine = -1
    peers = [r for r in range(local_world) if (nodes[r] if mine >= 0 else -1) == mine or mine < 0]  # ranks sharing this CPU pool
    # whole physical cores: group the pool's CPUs by their SMT sibling set
    cores, seen = [], set()
    for c in cpus:
        if c in seen:
            continue
        sib = _read(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list")
        grp = [x for x in (_parse_cpulist(sib) if sib else [c]) if x in set(cpus)] or [c]
        seen.update(grp)
        cores.append(grp)
    k = peers.index(local_rank)
    per = len(cores) // len(peers)
    if per < 1:  # fewer physical cores than ranks on this node: share, but say so
        return {"pinned": False, "why": f"{len(cores)} physical cores for {len(peers)} ranks", "numa_node": mine, "cpus": len(cpus)}
    take = cores[k * per:(k + 1) * per][:8]  # (a rank runs one launch thread: 8 cores are plenty, the rest stay free)
    mask = sorted(x for grp in take for x in grp)
    try:
        os.sched_setaffinity(0, mask)
    except OSError as exc:
        return {"pinned": False, "why": str(exc), "numa_node": mine}
    return {"pinned": True, "numa_node": mine, "physical_cores": len(take), "cpus": mask if len(mask) <= 16 else f"{mask[0]}-{mask[-1]} ({len(mask)})"}


HBM_REGIME_BATCH, HBM_REGIME_STEPS, HBM_REGIME_WARMUP, HBM_REGIME_TAPE = 262144, 300, 30, 8
API_LOOP_STEPS, API_LOOP_WARMUP = 2000, 200


def pmc_traffic(env_id, B, sha, sensor_range=0, observation_type=1, msg_bits=0):
    """Physical bytes per launch from the rocprofv3 --pmc passes (profiles/pmc_traffic.json, written by profiles/tools/sweep.sh
    and tied to the kernel sources by hash).  Returns (bytes or None, note or None)."""
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(pmc):
        return None, "profiles/pmc_traffic.json not found"
    try:
        rec = json.load(open(pmc))
        key = f"{env_id}:{B}" + (f":r{sensor_range}" if sensor_range else "") \
              + (f":obs{observation_type}" if observation_type != 1 else "") + (f":m{msg_bits}" if msg_bits else "")
        ent = rec.get("entries", {}).get(key)
        if ent is None:
            return None, f"no PMC traffic recorded for {key}"
        if rec.get("kernel_sources_sha") != sha:
            return None, (f"stale: profiles/pmc_traffic.json was measured at kernel sources {rec.get('kernel_sources_sha')}, "
                          f"this build is {sha}; re-run profiles/tools/sweep.sh")
        return int(ent["bytes_per_launch"]), None
    except Exception as exc:  # noqa: BLE001
        return None, f"unreadable pmc_traffic.json: {exc}"


def hbm_regime_leg(torch, rware_amd, local_rank, env_id, sha):
    """The same kernel at a batch whose per-step traffic no longer fits the 256 MiB Infinity Cache ("also run
    a cache-exceeding batch for the roofline claim"): small-4ag x 262144 envs = 298 MB of observations per step.  HIP events on
    the first / last launch of the timed region, like the headline leg."""
    B, K, W, TS = HBM_REGIME_BATCH, HBM_REGIME_STEPS, HBM_REGIME_WARMUP, HBM_REGIME_TAPE
    kw = rware_amd.env_kwargs(env_id)
    N = kw["n_agents"]
    env = rware_amd.WarehouseVecEnv(B, devices=[local_rank], **kw)
    try:
        eng = env.engines[0]
        eng.reset(seeds=rware_amd.shard_seeds(0, B))
        acts = np.random.default_rng(777).integers(0, 5, size=(TS, B, N), dtype=np.int32)
        tape = torch.from_numpy(acts).to(f"cuda:{local_rank}")
        torch.cuda.synchronize()
        eng.step_tape_device_timed(tape.data_ptr(), TS, 0, W, 0, 1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.step_tape_device_timed(tape.data_ptr(), TS, W % TS, K, 0, 1)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        k_ms = eng.event_elapsed_ms(0, 1) / K
        eng.sync()
        try:
            floor_ms = eng.debug_store_floor(100)
        except Exception:  # noqa: BLE001
            floor_ms = None
        info = eng.info
        a_bytes = int(info.algorithmic_bytes_per_env_step) * B
        e_bytes = int(info.engine_bytes_per_env_step) * B
        traffic, note = pmc_traffic(env_id, B, sha)
        out = {
            "workload": f"{env_id} batch={B} envs on one GPU (observations {B * N * int(info.obs_length) * 4 / 1e6:.0f} MB per step: "
                        "past the 256 MiB Infinity Cache), same per-step launches as `value`",
            "steps": K, "warmup": W, "ms_per_step": wall / K * 1e3, "kernel_ms_per_launch": k_ms,
            "value": B * N * K / wall, "unit": "agent-steps/s",
            # (`achieved` / `frac`: physical bytes / time; `*_algorithmic`: SURVEY.md §8(d)'s bytes, a work rate — see roofline_fields)
            **roofline_fields(a_bytes, e_bytes, traffic, k_ms), "unit_bw": "GB/s",
            "store_only_kernel_ms_per_launch": floor_ms, "frac_of_store_only_kernel": floor_fraction(floor_ms, k_ms),
            "envs_per_workgroup": int(info.envs_per_workgroup), "kernel_specialised": bool(info.specialised),
            "start_stagger_ns_per_slot": 10 * int(info.stagger_ticks),   # launches of two or more rounds of workgroups (rw_info.stagger_ticks)
            "wave_priority": int(info.wave_priority),   # 1: the chain in front of the first store runs at raised wavefront priority (rw_info.wave_priority)
            "observation_stores": "non-temporal" if int(info.obs_stores_stream) else "cached",
        }
        if note:
            out["traffic_note"] = note
        return out
    finally:
        env.close()


def api_closed_loop_leg(torch, rware_amd, local_rank, env_id, B):
    """End-to-end API throughput: what a training loop gets from `WarehouseVecEnv(output="torch").step(
    cuda_actions)` — Python -> ctypes -> launch per step on torch's current stream, results as zero-copy torch views, NO sync
    inside the loop (the policy's next op is ordered behind the step by the stream), one sync at the end."""
    kw = rware_amd.env_kwargs(env_id)
    N = kw["n_agents"]
    dev = f"cuda:{local_rank}"
    env = rware_amd.WarehouseVecEnv(B, devices=[local_rank], output="torch", **kw)
    try:
        env.reset(seed=0)
        acts = torch.from_numpy(np.random.default_rng(4242).integers(0, 5, size=(64, B, N), dtype=np.int32)).to(dev)
        slices = [acts[t] for t in range(64)]  # what a policy hands over each step: one (B, N) int32 CUDA tensor
        for t in range(API_LOOP_WARMUP):
            env.step(slices[t % 64])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in range(API_LOOP_STEPS):
            obs, rew, term, trunc, _ = env.step(slices[t % 64])
        t_issue = time.perf_counter() - t0
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        env.sync()
        assert obs.is_cuda and term.dtype == torch.bool
        return {
            "what": "WarehouseVecEnv(output='torch').step(cuda int32 actions) from Python, one call per step, no sync inside the "
                    "loop, one device sync at the end",
            "steps": API_LOOP_STEPS, "warmup": API_LOOP_WARMUP, "us_per_step": wall / API_LOOP_STEPS * 1e6,
            "host_issue_us_per_step": t_issue / API_LOOP_STEPS * 1e6,
            "value": B * N * API_LOOP_STEPS / wall, "unit": "agent-steps/s", "envs": B,
        }
    finally:
        env.close()


def two_pipelines_leg(torch, rware_amd, local_rank, env_id, B):
    """The batch as TWO independent sub-batches (`rware_amd.make_pipelines(B, 2)`: one engine and one stream each), stepped
    concurrently — what a trainer with double-buffered sampling does (policy on half A while half B steps).  The workgroups of one
    launch run their load / agent / store phases in lock-step; two free-running launches of half the size drift out of phase and
    fill each other's idle phases (profiles/EXPERIMENTS.md §9-10).  Same device action tapes, K per-step launches per half from a
    launcher thread each (rw_step_tape_device releases the GIL), wall clock from the first enqueue to the last sync; the one-engine
    number next to it is measured the same way, here, so the two are comparable.  Then the same through the Python API from ONE host
    thread (`env.step(cuda_actions)` per sub-batch, no sync inside the loop).  Reported beside the headline, never as `value`."""
    import threading
    dev = f"cuda:{local_rank}"
    out = {"submit": "rware_amd.make_pipelines(B, 2): two engines on one device, B / 2 envs each, own streams; `tape`: K per-step launches per "
                     "half, one launcher thread each; `api`: env.step(cuda_actions) per half, alternating, one host thread",
           "steps": 2000, "tasks": {}}
    for task, b, extra in ((env_id, B, {}), ("rware-small-10ag-v1", 16384, {}), ("rware-large-16ag-v1", 16384, {}),
                           ("rware-large-16ag-v1", 16384, {"sensor_range": 2})):   # (the last one: BASELINE config 5's shard)
        kw = dict(rware_amd.env_kwargs(task), **extra)
        N, K, KA, TS = kw["n_agents"], out["steps"], 1000, 64
        acts = torch.from_numpy(np.random.default_rng(7).integers(0, 5, size=(TS, b, N), dtype=np.int32)).to(dev)
        row = {}
        for M in (1, 2):
            if M == 1:
                envs = [rware_amd.WarehouseVecEnv(b, devices=[local_rank], output="torch", **kw)]
                envs[0].reset(seed=0)
                bounds = [(0, b)]
            else:
                pipes = rware_amd.make_pipelines(b, 2, device=local_rank, **kw)
                for p_ in pipes:
                    p_.reset(seed=0)
                envs, bounds = [p_.env for p_ in pipes], [(p_.lo, p_.hi) for p_ in pipes]
            try:
                engs = [e.engines[0] for e in envs]
                tapes = [acts[:, lo:hi].contiguous() for lo, hi in bounds]
                torch.cuda.synchronize()

                def run(k, n):
                    engs[k].step_tape_device(tapes[k].data_ptr(), TS, 0, n)
                    engs[k].sync()
                for k in range(M):
                    run(k, 200)
                best = None
                for _ in range(3):
                    th = [threading.Thread(target=run, args=(k, K)) for k in range(M)]
                    t0 = time.perf_counter()
                    for t in th:
                        t.start()
                    for t in th:
                        t.join()
                    wall = time.perf_counter() - t0
                    best = wall if best is None else min(best, wall)
                name = "one_engine" if M == 1 else "two_pipelines"
                row[name] = {"tape_us_per_step": best / K * 1e6, "value": b * N * K / best, "unit": "agent-steps/s"}
                sl = [[tp[t] for t in range(TS)] for tp in tapes]
                for t in range(100):
                    for k, e in enumerate(envs):
                        e.step(sl[k][t % TS])
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for t in range(KA):
                    for k, e in enumerate(envs):
                        e.step(sl[k][t % TS])
                torch.cuda.synchronize()
                row[name]["api_us_per_step"] = (time.perf_counter() - t0) / KA * 1e6
            finally:
                for e in envs:
                    e.close()
        row["speedup"] = row["one_engine"]["tape_us_per_step"] / row["two_pipelines"]["tape_us_per_step"]
        out["tasks"][f"{task} x {b}" + "".join(f" {k}={v}" for k, v in extra.items())] = row
    return out


def submit_modes_leg(torch, rware_amd, rank, local_rank, kw, B, tape_ptr, args, dist):
    """Per rank: (1) host time per issued launch — n launches enqueued through the library's native loop on an idle stream, clock
    stopped when the call returns (nothing waited for); (2) the per-step launches of the timed region captured into a HIP graph
    (min(steps, one tape pass) of them) and replayed: one host call per replay.  A second engine of the same shape on a stream
    torch can capture; all ranks run this side by side (they arrive together from the timed region's closing fence)."""
    G = max(1, min(args.steps, TAPE_STEPS))
    gs = torch.cuda.Stream(device=local_rank)
    with torch.cuda.stream(gs):
        env2 = rware_amd.WarehouseVecEnv(B, devices=[local_rank], envs_per_workgroup=args.envs_per_wg,
                                         threads_per_workgroup=args.threads_per_wg, output="torch", **kw)
    try:
        e2 = env2.engines[0]
        e2.reset(seeds=rware_amd.shard_seeds(rank, B))
        with torch.cuda.stream(gs):
            e2.step_tape_device(tape_ptr, TAPE_STEPS, 0, 64)
            torch.cuda.synchronize()
            # (no barrier in here: the ranks arrive together from the timed region's closing fence, and a rank that failed in this
            #  side measurement must not leave the others waiting at a rendezvous)
            n = 256
            t0 = time.perf_counter()
            e2.step_tape_device(tape_ptr, TAPE_STEPS, 0, n)
            issue = time.perf_counter() - t0
            torch.cuda.synchronize()
            native = time.perf_counter() - t0
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=gs):
                e2.step_tape_device(tape_ptr, TAPE_STEPS, 0, G)
            torch.cuda.synchronize()
            graph.replay()
            torch.cuda.synchronize()
            reps = max(1, 512 // G)
            t0 = time.perf_counter()
            for _ in range(reps):
                graph.replay()
            torch.cuda.synchronize()
            g = time.perf_counter() - t0
        e2.sync()
        return {"host_issue_us_per_step": issue / n * 1e6, "native_ms_per_step_256": native / n * 1e3,
                "graph_ms_per_step": g / (reps * G) * 1e3, "graph_launches_per_replay": G}
    except Exception as exc:  # noqa: BLE001  (a side measurement: never at the price of the line)
        return {"submit_modes_error": repr(exc)}
    finally:
        env2.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20000)
    ap.add_argument("--warmup", type=int, default=500)
    ap.add_argument("--batch", type=int, default=BATCH_PER_GPU, help="envs per GPU (headline: 16384)")
    ap.add_argument("--env-id", default=ENV_ID)
    ap.add_argument("--sensor-range", type=int, default=0, help="override sensor_range (BASELINE config 5 uses 2)")
    ap.add_argument("--observation-type", type=int, default=1, help="1 FLATTENED (headline), 2 IMAGE, 3 IMAGE_DICT")
    ap.add_argument("--msg-bits", type=int, default=0)
    ap.add_argument("--envs-per-wg", type=int, default=0)
    ap.add_argument("--threads-per-wg", type=int, default=0)
    ap.add_argument("--many", type=int, default=0,
                    help="fused rollout: submit steps in chunks of this many through rw_step_many_device (one launch per chunk, "
                         "env chunk resident in LDS across the steps; open-loop)")
    ap.add_argument("--submit", choices=["auto", "native", "python", "graph"], default="auto",
                    help="who issues the per-step launches: the library's loop over the device action tape (rw_step_tape_device, "
                         "default), one Python -> ctypes rw_step_device call per step, or the same launches captured once in a "
                         "HIP graph (one whole pass over the action tape) and replayed — for rocprofv3 traces of the small "
                         "kernels, where the profiled host cannot issue single launches fast enough; the launches are identical.  "
                         "auto = native at EVERY N (the N = 1 number and the N > 1 numbers of a scaling curve then take the same "
                         "path); every rank also reports, beside it, the same launches replayed from a HIP graph and the host time "
                         "per issued launch (`ranks[i].graph_ms_per_step`, `.host_issue_us_per_step`) — SURVEY.md §8(e)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fused-extra", action="store_true", help="skip the extra fused-rollout measurement (profiling runs)")
    ap.add_argument("--no-sustained", action="store_true", help="skip the fixed 2000-step sustained leg (profiling runs)")
    ap.add_argument("--no-soak", action="store_true", help="skip the >= 3 s soak leg (profiling runs)")
    ap.add_argument("--no-hbm-regime", action="store_true", help="skip the cache-exceeding leg (small-4ag x 262144 envs)")
    ap.add_argument("--no-api-loop", action="store_true", help="skip the Python closed-loop API leg")
    ap.add_argument("--no-submit-modes", action="store_true", help="skip the per-rank host-issue / HIP-graph side measurement")
    args = ap.parse_args()

    if args.submit == "auto":
        args.submit = "native"  # (the same submission path at every N; the graph path is measured beside it, per rank)
    children = []
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        import torch  # (before spawning: a rank without a device would leave the others waiting at the rendezvous)

        n_vis = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if n_vis < args.gpus and os.environ.get("RWARE_BENCH_SHARE_GPU") != "1":
            raise SystemExit(f"--gpus {args.gpus} but only {n_vis} HIP device(s) visible")
        children = spawn_ranks(args.gpus)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch

    import rware_amd

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the engine has no CPU fallback")
    n_dev = torch.cuda.device_count()
    if os.environ.get("RWARE_BENCH_SHARE_GPU") == "1":  # test hook: several ranks on one device (1-GPU test box)
        local_rank %= n_dev
    elif local_rank >= n_dev:
        raise SystemExit(f"rank {rank}: --gpus {args.gpus} but only {n_dev} HIP device(s) visible")
    torch.cuda.set_device(local_rank)
    share = os.environ.get("RWARE_BENCH_SHARE_GPU") == "1"
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    try:
        placement = pin_rank(torch, int(os.environ.get("LOCAL_RANK", "0")), local_world, (lambda r: r % n_dev) if share else (lambda r: min(r, n_dev - 1)))
    except Exception as exc:  # noqa: BLE001  (placement is an optimisation: never let it take a rank down)
        placement = {"pinned": False, "why": f"pin_rank failed: {exc!r}"}
    dist = None
    if world > 1:
        import torch.distributed as dist

        # barrier + MAX of three floats: CPU tensors over gloo — the data path has no collective, so no RCCL at all
        import datetime

        dist.init_process_group(backend="gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=300))

    kw = rware_amd.env_kwargs(args.env_id)
    if args.sensor_range:
        kw["sensor_range"] = args.sensor_range
    if args.observation_type != 1:
        kw["observation_type"] = args.observation_type
    if args.msg_bits:
        kw["msg_bits"] = args.msg_bits
    B, N, AM = args.batch, kw["n_agents"], 1 + args.msg_bits
    gstream = None
    if args.submit == "graph":  # the engine has to sit on a stream torch can capture: its "torch output" mode does that
        gstream = torch.cuda.Stream(device=local_rank)
        with torch.cuda.stream(gstream):
            env = rware_amd.WarehouseVecEnv(B, devices=[local_rank], envs_per_workgroup=args.envs_per_wg,
                                            threads_per_workgroup=args.threads_per_wg, output="torch", **kw)
    else:
        env = rware_amd.WarehouseVecEnv(B, devices=[local_rank], envs_per_workgroup=args.envs_per_wg,
                                        threads_per_workgroup=args.threads_per_wg, **kw)
    eng = env.engines[0]
    info = eng.info
    # env i of rank r is global env r*B + i -> SeedSequence(r*B + i): results independent of the GPU count
    eng.reset(seeds=rware_amd.shard_seeds(rank, B))
    acts = np.random.default_rng(12345 + rank).integers(0, 5, size=(TAPE_STEPS, B, N, AM), dtype=np.int32)
    if AM > 1:
        acts[..., 1:] &= 1  # message bits
    tape = torch.from_numpy(acts).to(f"cuda:{local_rank}")
    base, stride = tape.data_ptr(), B * N * AM * 4

    graph, G = None, max(1, min(args.steps, TAPE_STEPS))
    if gstream is not None:  # G = min(steps, one pass over the tape) per-step launches, captured once: a short timed region
        torch.cuda.synchronize()  # (the driver's --steps 20) replays it too instead of falling back to plain launches
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=gstream):
            eng.step_tape_device(base, TAPE_STEPS, 0, G)
        torch.cuda.synchronize()

    def run(n, t_start, many=args.many):
        if graph is not None and many == 0:
            with torch.cuda.stream(gstream):
                for _ in range(n // G):  # (the graph always reads tape rows 0 .. G - 1: which random actions a step gets is immaterial)
                    graph.replay()
                if n % G:   # the tail: plain launches on the same stream
                    eng.step_tape_device(base, TAPE_STEPS, 0, n % G)
            return
        if many > 0:
            t = t_start
            while t < t_start + n:
                c = min(many, t_start + n - t, TAPE_STEPS - (t % TAPE_STEPS))
                eng.step_many_device(base + (t % TAPE_STEPS) * stride, c)
                t += c
        elif args.submit == "python":
            for t in range(t_start, t_start + n):
                eng.step_device(base + (t % TAPE_STEPS) * stride)
        else:  # the same n launches (one rw_step_device per step), issued by the library's native loop
            eng.step_tape_device(base, TAPE_STEPS, t_start % TAPE_STEPS, n)

    def fence():  # barrier + device sync, both sides of every timed region
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def timed(n, t_start, many=args.many):
        fence()
        t0 = time.perf_counter()
        if many == 0 and args.submit == "native" and n > 0:
            # the HIP events ride on the first / last launch (their own start / end timestamps): two marker packets
            # around a 20-launch region cost ~9 us of its ~170 (profiles/r02_k20_probe.txt)
            eng.step_tape_device_timed(base, TAPE_STEPS, t_start % TAPE_STEPS, n, 0, 1)
        else:
            eng.event_record(0)
            run(n, t_start, many)
            eng.event_record(1)
        fence()
        dt = time.perf_counter() - t0
        return dt, eng.event_elapsed_ms(0, 1) / max(n, 1)  # wall seconds; HIP-event ms per step on the engine's stream

    t_next = 0
    if args.warmup > 0:  # the W untimed steps take the same host path as the timed region (same calls, same brackets)
        timed(args.warmup, t_next)
        t_next += args.warmup
    elapsed, kernel_ms = timed(args.steps, t_next)
    t_next += args.steps
    eng.sync()  # surfaces a sticky device-side error (invalid action) outside the timed region

    sus = None
    if not args.no_sustained:
        run(SUSTAINED_WARMUP, t_next)
        t_next += SUSTAINED_WARMUP
        sus = timed(SUSTAINED_STEPS, t_next)
        t_next += SUSTAINED_STEPS

    # Soak: the same per-step launches back to back for >= SOAK_SECONDS — long enough for a utilisation sampler outside this process
    # (the driver's gpu_busy probe sees < 1 s of GPU work in the legs above) and for the clocks to settle.  Rank-local, no barrier.
    soak = None
    if not args.no_soak and args.many == 0:
        per_step = (sus[0] / SUSTAINED_STEPS) if sus else (elapsed / args.steps)
        n_soak = int(SOAK_SECONDS / max(per_step, 1e-7)) + 1
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.step_tape_device_timed(base, TAPE_STEPS, t_next % TAPE_STEPS, n_soak, 0, 1)
        torch.cuda.synchronize()
        soak = (time.perf_counter() - t0, eng.event_elapsed_ms(0, 1) / n_soak, n_soak)
        t_next += n_soak
        eng.sync()

    # Extra (reported beside the headline, never as `value`): the same K steps through the fused rollout
    # API, rw_step_many_device — one launch per 64 steps, env chunk resident in LDS (open-loop).
    fused = None
    if not args.many and not args.no_fused_extra:
        run(64, t_next, many=64)
        t_next += 64
        fused = timed(args.steps, t_next, many=64)
    eng.sync()

    # The practical floor beside the 8 TB/s one: a kernel that only WRITES this step's observations (same geometry, same store
    # instruction), launched back to back like the step kernel (rw_debug_store_floor)
    try:
        floor_ms = eng.debug_store_floor(2000 if B * N <= 1 << 18 else 200) if args.many == 0 else None
    except Exception:  # noqa: BLE001  (a side measurement)
        floor_ms = None
    # Every rank, at every N (N = 1 included): what a launch costs this rank's host thread, and the same launches replayed from a HIP
    # graph — so that a scaling curve can be read for host-side effects (a straggler rank, launch-rate contention) rank by rank
    modes = submit_modes_leg(torch, rware_amd, rank, local_rank, kw, B, base, args, dist) if (args.many == 0 and not args.no_submit_modes) else {}
    vals = [elapsed, kernel_ms, sus[0] if sus else 0.0, sus[1] if sus else 0.0, fused[0] if fused else 0.0]
    if dist is not None:
        tt = torch.tensor(vals, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        vals = [float(v) for v in tt]
    mine = {"rank": rank, "device": local_rank, "ms_per_step": elapsed / args.steps * 1e3, "kernel_ms_per_launch": kernel_ms,
            "sustained_ms_per_step": (sus[0] / SUSTAINED_STEPS * 1e3) if sus else None, "placement": placement, **modes}
    per_rank = [mine]
    if dist is not None:  # (objects over gloo: still nothing on the data path)
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
    elapsed, kernel_ms, sus_s, sus_kernel_ms, fused_s = vals

    if rank == 0:
        a_bytes = int(info.algorithmic_bytes_per_env_step)
        per_launch = a_bytes * B
        e_launch = int(info.engine_bytes_per_env_step) * B  # what this engine's layout has to move per launch (rw_info)
        k_ms = kernel_ms  # HIP-event time per step (== per launch unless --many fuses several steps into one launch)
        sha = kernel_sources_sha()
        traffic, traffic_note = pmc_traffic(args.env_id, B, sha, args.sensor_range, args.observation_type, args.msg_bits)
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
                "parallelism": f"env-shard x{world} (no collective, no RCCL; gloo barrier + MAX of the timings only); each rank pinned to "
                               f"whole physical cores of its GPU's NUMA node ({sum(1 for r in per_rank if r['placement'].get('pinned'))}/{world} pinned, see `ranks`)",
                "submit": f"rw_step_many_device x{args.many} (fused rollout, one launch per chunk)" if args.many
                          else "one rw_step_device launch per step (closed-loop capable kernel), issued by "
                               + ("rw_step_tape_device's native loop over the device action tape" if args.submit == "native"
                                  else "a HIP graph holding one pass over the action tape (captured rw_step_tape_device), replayed"
                                  if args.submit == "graph" else "one Python/ctypes call per step"),
                "envs_per_workgroup": int(info.envs_per_workgroup), "threads_per_workgroup": int(info.threads_per_workgroup),
                "start_stagger_ns_per_slot": 10 * int(info.stagger_ticks),   # 0: the launch is resident at once (no stagger)
                "wave_priority": int(info.wave_priority),
                "kernel_specialised": bool(info.specialised), "kernel_build_kind": int(info.build_kind),
                "observation_stores": "non-temporal (the engine's default rule for this shape; rw_stream_flags / obs_stores= overrides)"
                                      if int(info.obs_stores_stream) else "cached",
                "device": info.device_name.decode(), "arch": info.arch_name.decode(),
            },
            "roofline": {
                # `achieved` / `frac`: the PHYSICAL bandwidth of the step kernel — PMC bytes of a recorded rocprofv3 pass on these kernel
                # sources (`traffic`), else the bytes this engine's layout has to move (rw_info.engine_bytes_per_env_step) — over the
                # kernel time THIS run measured: a fraction of the 8 TB/s peak, <= 1.  `achieved_algorithmic` / `frac_algorithmic`:
                # SURVEY.md §8(d)'s bytes (two int32 grid layers and five int32 agent fields per env, which the engine replaces by a
                # 1-byte shelf layer and one dword per agent) over the same time: a work rate in that unit, it may exceed 1.
                "bound": "hbm", "unit": "GB/s", **roofline_fields(per_launch, e_launch, traffic, k_ms),
                "note": "achieved / frac = physical bytes per launch (basis: PMC traffic of the recorded rocprofv3 pass when it belongs to "
                        "these kernel sources, else the engine-layout bytes) / kernel time / 8 TB/s; regime says whether those bytes "
                        "fit the 256 MiB Infinity Cache (then the figure is a cache bandwidth, not an HBM one); *_algorithmic prices "
                        "SURVEY.md §8(d)'s bytes, which the engine does not move",
                "kernel": "rw::rware_step_kernel", "kernel_ms_per_launch": k_ms,
                # a kernel that does nothing but write this step's observations, same launch geometry and store instruction, launched
                # back to back: what ANY kernel producing these observations takes at the least on this device, and the step kernel's
                # time as a multiple of it (the simulation, the window gather and the state traffic are what is on top)
                "store_only_kernel_ms_per_launch": floor_ms, "frac_of_store_only_kernel": floor_fraction(floor_ms, k_ms),
                "kernel_sources_sha": sha,
            },
        }
        out["ranks"] = per_rank  # a straggler (or a badly placed rank) shows here: `value` uses the MAX over ranks
        if traffic_note:
            out["roofline"]["traffic_note"] = traffic_note
        if sus_s:
            s_rf = roofline_fields(per_launch, e_launch, traffic, sus_kernel_ms)
            out["sustained"] = {
                "value": world * B * N * SUSTAINED_STEPS / sus_s, "unit": "agent-steps/s",
                "steps": SUSTAINED_STEPS, "warmup": SUSTAINED_WARMUP, "ms_per_step": sus_s / SUSTAINED_STEPS * 1e3,
                "kernel_ms_per_launch": sus_kernel_ms, "roofline_achieved": s_rf["achieved"], "roofline_frac": s_rf["frac"],
                "roofline_basis": s_rf["basis"], "roofline_regime": s_rf["regime"],
                "roofline_frac_engine": s_rf["frac_engine"], "roofline_frac_physical": s_rf["frac_physical"],
                "roofline_achieved_algorithmic": s_rf["achieved_algorithmic"], "roofline_frac_algorithmic": s_rf["frac_algorithmic"],
                "what": "same launches as `value`, fixed length (SURVEY.md §8(d): 2000 steps after 100 warm-up, spans 4 mass resets)",
            }
        if soak:
            k_rf = roofline_fields(per_launch, e_launch, traffic, soak[1])
            out["soak"] = {
                "what": "the same per-step launches as `value`, back to back for >= 3 s on rank 0 (steps x ms_per_step = wall_s: check it against "
                        "the driver's clock; long enough for an outside GPU-utilisation sampler to see the device busy)",
                "steps": soak[2], "ms_per_step": soak[0] / soak[2] * 1e3, "wall_s": soak[0], "kernel_ms_per_launch": soak[1],
                "value": B * N * soak[2] / soak[0], "unit": "agent-steps/s (this rank)", "roofline_frac": k_rf["frac"], "roofline_basis": k_rf["basis"],
            }
        if fused_s:
            out["fused_rollout"] = {
                "value": world * B * N * args.steps / fused_s, "unit": "agent-steps/s",
                "ms_per_step": fused_s / args.steps * 1e3,
                "submit": "rw_step_many_device x64: one launch per 64 steps, env chunk resident in LDS across steps "
                          "(open-loop rollout from the same device action tape; identical results)",
            }
        extra = world == 1 and args.env_id == ENV_ID and args.many == 0 and args.observation_type == 1 and not args.msg_bits and not args.sensor_range
        if extra and not (args.no_hbm_regime and args.no_api_loop):
            env.close()  # one engine at a time on the device
            if not args.no_api_loop:
                out["api_closed_loop"] = api_closed_loop_leg(torch, rware_amd, local_rank, args.env_id, B)
                out["api_closed_loop"]["vs_native_loop"] = out["api_closed_loop"]["us_per_step"] / (out["ms_per_step"] * 1e3)
            if not args.no_hbm_regime:
                out["hbm_regime"] = hbm_regime_leg(torch, rware_amd, local_rank, args.env_id, sha)
            if not args.no_api_loop:
                try:  # (an extra: never at the price of the line)
                    out["two_pipelines"] = two_pipelines_leg(torch, rware_amd, local_rank, args.env_id, B)
                except Exception as exc:  # noqa: BLE001
                    out["two_pipelines"] = {"error": repr(exc)}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.env_id)
        print(json.dumps(out), flush=True)
    env.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    rc = 0
    for c in children:
        rc |= c.wait(timeout=120)
    if rc:
        raise SystemExit(f"a spawned rank exited with status {rc}")


if __name__ == "__main__":
    main()
