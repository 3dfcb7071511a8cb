"""Event counters (rw_stream_flags RW_STATS_ON; SURVEY.md §5 "metrics"): per env, running totals of shelf deliveries
(rware/warehouse.py:907-917) and of FORWARD requests the step turned into NOOP (:836-846, :871-876).

The pin is the reference itself: tests/golden/events/events.npz holds both counts per step and env, read off the UNMODIFIED
reference's objects while it replayed the golden traces (tests/golden/generate_events.py; oracle/ref_runner.py ref_step_events).
  - the C oracle against that fixture (every trace, every step) — and live against /root/reference in test_oracle_vs_reference.py;
  - the product sources on host threads (tests/emu) against the fixture and the oracle: every autoreset mode, fused rollouts,
    snapshots, the off switch;
  - the gfx950 library (-m gpu) against the fixture on every FLATTENED trace in full and against the oracle at larger batches.
"""
import json
import os

import numpy as np
import pytest

import golden_util as gu
from rware_oracle import OracleVecEnv

import rware_amd

EVENTS = np.load(os.path.join(gu.GOLDEN_DIR, "events", "events.npz"))


def fixture_totals(name):
    """cumulative (deliveries, failed moves) [T][E] of a golden trace, as the reference counted them"""
    return (np.cumsum(EVENTS[name + "/deliveries"].astype(np.int64), axis=0),
            np.cumsum(EVENTS[name + "/failed"].astype(np.int64), axis=0))


def test_events_fixture_covers_every_golden_trace():
    meta = json.loads(str(EVENTS["meta"]))
    assert set(meta["totals"]) == set(gu.fixture_names())
    assert sum(v["deliveries"] for v in meta["totals"].values()) == 731 and sum(v["failed"] for v in meta["totals"].values()) == 10205
    for name in gu.fixture_names():
        m, _ = gu.load_fixture(name)
        assert EVENTS[name + "/deliveries"].shape == (m["T"], m["E"]) == EVENTS[name + "/failed"].shape


@pytest.mark.parametrize("name", gu.fixture_names())
def test_oracle_event_counts_match_the_reference(name):
    meta, z = gu.load_fixture(name)
    orc = OracleVecEnv(meta["E"], **gu.ctor_kwargs(meta))
    orc.reset(seed=meta["seed"])
    want_d, want_f = fixture_totals(name)
    for t in range(meta["T"]):
        orc.step_autoreset(z["actions"][t].astype(np.int32), "next_step")
        assert np.array_equal(orc.stat_deliveries, want_d[t]), ("deliveries", t)
        assert np.array_equal(orc.stat_failed_moves, want_f[t]), ("failed moves", t)


# ------------------------------------------------------------------------------------------------------------------------------
# the product, either library: `lib` is None for the gfx950 build (GPU tests) or the host-thread emulation build (CPU suite)
# ------------------------------------------------------------------------------------------------------------------------------
def replay_trace_with_counters(name, lib, steps, tile=1, want_build=None, **geom):
    from engine_backend import EngineBackend

    meta, z = gu.load_fixture(name)
    be = EngineBackend(meta["E"], library=lib, tile=tile, stats=True, **geom, **gu.ctor_kwargs(meta))
    info = be.env.engines[0].info
    assert info.stats == 1
    if want_build == "jit":   # a run-time compiled exact-shape build carrying the counting code (the ahead-of-time ones do not)
        assert info.jit in (1, 2) and info.specialised == 1 and info.build_kind == 1, be.env.engines[0].jit_log()
    be.reset(seed=meta["seed"])
    want_d, want_f = fixture_totals(name)
    T = min(steps or meta["T"], meta["T"])
    for t in range(T):
        _, rew, _ = be.step_autoreset(z["actions"][t].astype(np.int32), "next_step")
        assert np.array_equal(rew, z["rewards"][t]), t
        c = be.env.event_counters()
        assert np.array_equal(be._first(c["deliveries"]), want_d[t]), ("deliveries", name, t)
        assert np.array_equal(be._first(c["failed_moves"]), want_f[t]), ("failed moves", name, t)
    be.env.close()
    return T


def check_against_oracle(lib, env_id, extra, B, T, mode, geom=(0, 0), p=(.1, .55, .1, .1, .15), seed=17, jit=None):
    kw = rware_amd.env_kwargs(env_id)
    kw.update(extra)
    kw["reward_type"] = rware_amd.enums.enum_value(kw["reward_type"])
    env = rware_amd.WarehouseVecEnv(B, autoreset_mode=mode, library=lib, stats=True, envs_per_workgroup=geom[0],
                                    threads_per_workgroup=geom[1], jit=jit, **kw)
    assert not jit or env.engines[0].info.jit in (1, 2), env.engines[0].jit_log()
    orc = OracleVecEnv(B, **kw)
    assert np.array_equal(env.reset(seed=seed)[0], orc.reset(seed=seed))
    rng = np.random.default_rng(seed)
    M = kw.get("msg_bits", 0)
    for t in range(T):
        a = rng.choice(5, size=(B, kw["n_agents"]), p=list(p)).astype(np.int32)
        if M:
            a = np.concatenate([a[..., None], rng.integers(0, 2, size=(B, kw["n_agents"], M), dtype=np.int32)], axis=-1)
        _, rew, term, _, info = env.step(a)
        _, r2, d2 = orc.step_autoreset(a, mode)
        assert np.array_equal(rew, r2) and np.array_equal(term, d2.astype(bool)), t
        assert "deliveries" not in info and "failed_moves" not in info   # the reference's info is {} (:746-747): the counters are a method
        c = env.event_counters()
        assert np.array_equal(c["deliveries"], orc.stat_deliveries), ("deliveries", t)
        assert np.array_equal(c["failed_moves"], orc.stat_failed_moves), ("failed moves", t)
    totals = int(orc.stat_deliveries.sum()), int(orc.stat_failed_moves.sum())
    env.close()
    return totals


def check_rollout(lib, env_id, extra, B, T, mode, geom=(0, 0), jit=None):
    """fused rollouts (one launch, T steps) add the same events as T single steps, and the stepwise path carries on from there"""
    kw = rware_amd.env_kwargs(env_id)
    kw.update(extra)
    kw["reward_type"] = rware_amd.enums.enum_value(kw["reward_type"])
    env = rware_amd.WarehouseVecEnv(B, autoreset_mode=mode, library=lib, stats=True, envs_per_workgroup=geom[0],
                                    threads_per_workgroup=geom[1], jit=jit, **kw)
    assert not jit or env.engines[0].info.jit in (1, 2), env.engines[0].jit_log()
    orc = OracleVecEnv(B, **kw)
    env.reset(seed=5)
    orc.reset(seed=5)
    acts = np.random.default_rng(1).choice(5, size=(T + 3, B, kw["n_agents"]), p=[.1, .55, .1, .1, .15]).astype(np.int32)
    _, rew, _ = env.rollout(acts[:T], want_obs=False)
    for t in range(T):
        _, r2, _ = orc.step_autoreset(acts[t], mode)
        assert np.array_equal(rew[t], r2), t
    c = env.event_counters()
    assert np.array_equal(c["deliveries"], orc.stat_deliveries) and np.array_equal(c["failed_moves"], orc.stat_failed_moves)
    for t in range(T, T + 3):
        env.step(acts[t])
        orc.step_autoreset(acts[t], mode)
    c = env.event_counters()
    assert np.array_equal(c["deliveries"], orc.stat_deliveries) and np.array_equal(c["failed_moves"], orc.stat_failed_moves)
    assert int(orc.stat_failed_moves.sum()) > 0
    env.close()


def check_switch_snapshot_and_writes(lib):
    kw = rware_amd.env_kwargs("rware-small-4ag-v1")
    kw["max_steps"] = 12
    B = 16
    off = rware_amd.WarehouseVecEnv(B, library=lib, **kw)          # the default: no counters, no buffers
    assert off.engines[0].info.stats == 0 and off.engines[0].stats is False
    with pytest.raises(RuntimeError, match="stats=True"):
        off.event_counters()
    for name in ("stat_deliveries", "stat_failed_moves"):
        with pytest.raises(rware_amd._capi.EngineError):   # the buffers are zero bytes long: reading B values from them is refused
            off.engines[0].read(name)
    on = rware_amd.WarehouseVecEnv(B, library=lib, stats=True, **kw)
    o0, _ = off.reset(seed=8)
    o1, _ = on.reset(seed=8)
    assert np.array_equal(o0, o1)
    acts = np.random.default_rng(2).choice(5, size=(40, B, 4), p=[.05, .7, .1, .1, .05]).astype(np.int32)
    for t in range(15):   # counting changes nothing else (episodes end and autoreset on the way: max_steps 12)
        a, b = off.step(acts[t]), on.step(acts[t])
        assert all(np.array_equal(x, y) for x, y in zip(a[:4], b[:4])), t
    assert all(np.array_equal(v, on.get_state()[k]) for k, v in off.get_state().items())
    before = on.event_counters()
    assert before["failed_moves"].sum() > 0 and before["failed_moves"].dtype == np.int32
    snap = on.snapshot()                      # snapshots carry the totals
    for t in range(15, 25):
        on.step(acts[t])
    later = on.event_counters()
    assert (later["failed_moves"] >= before["failed_moves"]).all() and later["failed_moves"].sum() > before["failed_moves"].sum()
    on.restore(snap)
    again = on.event_counters()
    assert np.array_equal(again["deliveries"], before["deliveries"]) and np.array_equal(again["failed_moves"], before["failed_moves"])
    for t in range(15, 25):
        on.step(acts[t])
    replay = on.event_counters()
    assert np.array_equal(replay["deliveries"], later["deliveries"]) and np.array_equal(replay["failed_moves"], later["failed_moves"])
    on.free_snapshot(snap)
    on.reset(seed=8)                          # reset() does not touch them ...
    assert np.array_equal(on.event_counters()["failed_moves"], later["failed_moves"])
    on.set_state(refresh_obs=False, stat_failed_moves=np.zeros(B, np.int32), stat_deliveries=np.full(B, 7, np.int32))   # ... the caller does
    c = on.event_counters()
    assert not c["failed_moves"].any() and (c["deliveries"] == 7).all()
    off.close()
    on.close()


# ------------------------------------------------------------------------------------------------------------------------------
# CPU suite: the product sources on host threads
# ------------------------------------------------------------------------------------------------------------------------------
def _emu():
    from engine_backend import build_emu
    return build_emu()


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("name,geom,tile,steps", [
    ("tiny-2ag", (4, 64), 1, 150),                 # generic kernel (LDS agent phases)
    ("small-4ag", (0, 0), 4, 150),                 # exact-shape build, agent phases in registers
    ("small-8ag-global-inact", (0, 0), 16, 60),    # GLOBAL rewards, inactivity limit
    ("tiny-4ag-easy-twostage", (4, 128), 1, 120),  # TWO_STAGE
    ("small-19ag", (4, 64), 1, 60),                # generic kernel, 19 agents
    ("msg2-small-4ag", (0, 0), 4, 100),            # actions are [Action, bits...]: the action re-read strides over the message words
])
def test_emulated_engine_counts_what_the_reference_counts(name, geom, tile, steps):
    assert replay_trace_with_counters(name, _emu(), steps, tile=tile, envs_per_workgroup=geom[0], threads_per_workgroup=geom[1]) == steps


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("mode", ["next_step", "same_step", "disabled"])
@pytest.mark.parametrize("env_id,extra,B,geom", [
    ("rware-tiny-2ag-v1", {"max_steps": 14}, 7, (4, 64)),                                       # generic, ragged batch
    ("rware-small-4ag-v1", {"max_steps": 16, "max_inactivity_steps": 9}, 16, (0, 0)),           # exact
    ("rware-small-10ag-v1", {"max_steps": 12}, 8, (0, 0)),                                      # agent-count-static, per-cell agent phases
    ("rware-large-16ag-v1", {"sensor_range": 2, "max_steps": 10, "reward_type": 0}, 4, (0, 0)),  # BASELINE config 5's kernel
])
def test_emulated_engine_counters_match_oracle_in_every_autoreset_mode(env_id, extra, B, geom, mode):
    d, f = check_against_oracle(_emu(), env_id, extra, B, 40, mode, geom, p=(.05, .7, .1, .1, .05))
    assert f > 0


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("env_id,extra,B,geom,mode", [
    ("rware-small-4ag-v1", {"max_steps": 13}, 32, (0, 0), "next_step"),
    ("rware-tiny-2ag-v1", {"max_steps": 9}, 7, (4, 64), "same_step"),
    ("rware-small-10ag-v1", {"max_steps": 11}, 8, (0, 0), "disabled"),
])
def test_emulated_fused_rollout_counts_like_single_steps(env_id, extra, B, geom, mode):
    check_rollout(_emu(), env_id, extra, B, 30, mode, geom)


@pytest.mark.timeout(1500)
def test_emulated_pipelined_build_counts_too(monkeypatch):
    """the chunk-pipelined persistent flow (a `make PIPE=1` library; the emulation build carries it) with counters: the service wavefront
    counts per chunk, several chunks per workgroup"""
    monkeypatch.setenv("RWARE_PIPE_GRID", "2")
    kw = rware_amd.env_kwargs("rware-small-4ag-v1")
    kw["max_steps"] = 14
    kw["reward_type"] = rware_amd.enums.enum_value(kw["reward_type"])
    B = 96   # 6 chunks of 16 envs on 2 persistent workgroups
    env = rware_amd.WarehouseVecEnv(B, library=_emu(), stats=True, pipe=True, **kw)
    assert env.engines[0].info.pipe_workgroups == 2 and env.engines[0].info.stats == 1
    orc = OracleVecEnv(B, **kw)
    assert np.array_equal(env.reset(seed=12)[0], orc.reset(seed=12))
    rng = np.random.default_rng(12)
    for t in range(35):
        a = rng.choice(5, size=(B, 4), p=[.05, .7, .1, .1, .05]).astype(np.int32)
        _, rew, _, _, _ = env.step(a)
        _, r2, _ = orc.step_autoreset(a, "next_step")
        assert np.array_equal(rew, r2), t
        c = env.event_counters()
        assert np.array_equal(c["deliveries"], orc.stat_deliveries) and np.array_equal(c["failed_moves"], orc.stat_failed_moves), t
    assert orc.stat_failed_moves.sum() > 0
    env.close()


@pytest.mark.timeout(1500)
def test_emulated_sharded_env_gathers_its_counters():
    """an env sharded over several engines (devices=[...]): event_counters() is the concatenation in env order, equal to the unsharded oracle"""
    kw = rware_amd.env_kwargs("rware-tiny-2ag-v1")
    kw["max_steps"] = 12
    B = 24
    env = rware_amd.WarehouseVecEnv(B, library=_emu(), devices=[0, 0, 0], stats=True, **kw)
    assert len(env.engines) == 3
    orc = OracleVecEnv(B, **kw)
    assert np.array_equal(env.reset(seed=3)[0], orc.reset(seed=3))
    rng = np.random.default_rng(9)
    for t in range(30):
        a = rng.choice(5, size=(B, 2), p=[.05, .7, .1, .1, .05]).astype(np.int32)
        env.step(a)
        orc.step_autoreset(a, "next_step")
    c = env.event_counters()
    assert c["deliveries"].shape == (B,) and np.array_equal(c["deliveries"], orc.stat_deliveries)
    assert np.array_equal(c["failed_moves"], orc.stat_failed_moves) and orc.stat_failed_moves.sum() > 0
    env.close()


@pytest.mark.timeout(1500)
def test_emulated_counters_switch_snapshot_and_writes():
    check_switch_snapshot_and_writes(_emu())


# ------------------------------------------------------------------------------------------------------------------------------
# GPU suite: the gfx950 library, through the C-ABI
# ------------------------------------------------------------------------------------------------------------------------------
FLAT = [n for n in gu.fixture_names()]


@pytest.mark.gpu
@pytest.mark.parametrize("name", FLAT)
def test_engine_counts_what_the_reference_counts(name):
    """every golden trace in full (generic or run-time shapes as rw_create picks them for 2-4 envs), counters against the reference's"""
    meta, _ = gu.load_fixture(name)
    assert replay_trace_with_counters(name, None, None) == meta["T"]


@pytest.mark.gpu
@pytest.mark.parametrize("name,geom,tile", [
    ("small-4ag", (0, 0), 4), ("tiny-2ag", (0, 0), 4), ("medium-6ag-hard", (0, 0), 8), ("large-16ag-sr2", (0, 0), 4), ("large-16ag-sr2", (4, 256), 4),
    ("small-8ag-global-inact", (0, 0), 16), ("tiny-4ag-easy-twostage", (0, 0), 16), ("small-7ag-hard", (0, 0), 4), ("small-19ag", (0, 0), 4),
    ("msg2-small-4ag", (0, 0), 16), ("img-small-4ag-directional", (0, 0), 16), ("medium-2ag-easy", (32, 256), 8),
])
def test_runtime_compiled_exact_shape_builds_count_what_the_reference_counts(name, geom, tile):
    """the exact-shape builds an engine with counters runs on: compiled at construction with the counting code (hipRTC, RW_STATS_BUILD)"""
    meta, _ = gu.load_fixture(name)
    assert replay_trace_with_counters(name, None, None, tile=tile, want_build="jit", jit=True, envs_per_workgroup=geom[0],
                                      threads_per_workgroup=geom[1]) == meta["T"]


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["next_step", "same_step", "disabled"])
@pytest.mark.parametrize("env_id,extra,B,T,jit", [
    ("rware-small-4ag-v1", {"max_steps": 60}, 2048, 200, True),                            # run-time compiled exact-shape build
    ("rware-tiny-2ag-v1", {"max_steps": 50, "max_inactivity_steps": 30}, 1024, 160, None),  # generic kernel
    ("rware-medium-6ag-hard-v1", {"max_steps": 70, "reward_type": 2}, 1024, 160, True),
    ("rware-small-10ag-v1", {"max_steps": 40}, 512, 120, True),                            # per-cell agent phases
    ("rware-medium-13ag-v1", {"max_steps": 40, "reward_type": 0}, 512, 100, None),
    ("rware-large-16ag-v1", {"sensor_range": 2, "max_steps": 50}, 256, 120, True),         # BASELINE config 5's shape
    ("rware-small-19ag-v1", {"max_steps": 40}, 256, 100, True),
    ("rware-small-4ag-v1", {"max_steps": 40, "observation_type": 2}, 512, 100, None),
])
def test_engine_counters_match_oracle_in_every_autoreset_mode(env_id, extra, B, T, jit, mode):
    d, f = check_against_oracle(None, env_id, extra, B, T, mode, p=(.08, .62, .1, .1, .1), jit=jit)
    assert f > 0


@pytest.mark.gpu
@pytest.mark.parametrize("env_id,extra,B,mode", [
    ("rware-small-4ag-v1", {"max_steps": 23}, 1024, "next_step"),
    ("rware-small-10ag-v1", {"max_steps": 19}, 256, "same_step"),
    ("rware-large-16ag-v1", {"sensor_range": 2, "max_steps": 17}, 64, "disabled"),
])
def test_fused_rollout_counts_like_single_steps(env_id, extra, B, mode):
    check_rollout(None, env_id, extra, B, 48, mode, jit=(B >= 256))


@pytest.mark.gpu
def test_an_engine_with_counters_gets_a_kernel_that_counts(tmp_path, monkeypatch):
    """rw_create's choice: without counters the ahead-of-time exact-shape build; with counters a run-time compiled one from 4096 envs on
    (rw_jit_log says why), the generic kernel below that; the BASELINE launch without counters is the same kernel as ever"""
    monkeypatch.setenv("RWARE_JIT_CACHE", str(tmp_path))
    kw = rware_amd.env_kwargs("rware-small-4ag-v1")
    plain = rware_amd.WarehouseVecEnv(4096, **kw)
    i = plain.engines[0].info
    assert (i.stats, i.jit, i.build_kind, i.specialised) == (0, 0, 1, 1) and plain.engines[0].jit_log() == ""
    counted = rware_amd.WarehouseVecEnv(4096, stats=True, **kw)
    j = counted.engines[0].info
    assert (j.stats, j.jit, j.build_kind, j.specialised) == (1, 1, 1, 1), counted.engines[0].jit_log()
    assert "do not carry the counting code" in counted.engines[0].jit_log()
    again = rware_amd.WarehouseVecEnv(4096, stats=True, **kw)                    # ... from the disk cache the second time
    assert again.engines[0].info.jit == 2
    small = rware_amd.WarehouseVecEnv(64, stats=True, **kw)
    k = small.engines[0].info
    assert (k.stats, k.jit, k.build_kind, k.specialised) == (1, 0, 0, 0)
    o0, _ = plain.reset(seed=1)
    o1, _ = counted.reset(seed=1)
    assert np.array_equal(o0, o1)
    a = np.random.default_rng(0).integers(0, 5, size=(4096, 4)).astype(np.int32)
    for _ in range(5):
        r0, r1 = plain.step(a), counted.step(a)
        assert all(np.array_equal(x, y) for x, y in zip(r0[:4], r1[:4]))
    for e in (plain, counted, again, small):
        e.close()


@pytest.mark.gpu
def test_counters_switch_snapshot_and_writes():
    check_switch_snapshot_and_writes(None)


@pytest.mark.gpu
def test_counters_are_zero_copy_torch_tensors():
    """output="torch": the counters are device tensors on the engine's own buffers, current after every step without a copy"""
    import torch

    kw = rware_amd.env_kwargs("rware-small-4ag-v1")
    B, T = 1024, 24
    env = rware_amd.WarehouseVecEnv(B, output="torch", stats=True, **kw)
    orc = OracleVecEnv(B, **kw)
    env.reset(seed=31)
    orc.reset(seed=31)
    acts = np.random.default_rng(4).choice(5, size=(T, B, 4), p=[.05, .7, .1, .1, .05]).astype(np.int32)
    c = env.event_counters()
    assert c["failed_moves"].is_cuda and c["failed_moves"].dtype == torch.int32 and c["failed_moves"].shape == (B,)
    assert c["failed_moves"].data_ptr() == env.engines[0].device_array("stat_failed_moves").ptr
    tape = torch.from_numpy(acts).cuda()
    for t in range(T):
        env.step(tape[t])
        orc.step_autoreset(acts[t], "next_step")
    torch.cuda.synchronize()
    assert np.array_equal(c["deliveries"].cpu().numpy(), orc.stat_deliveries)      # the same tensors, now current
    assert np.array_equal(c["failed_moves"].cpu().numpy(), orc.stat_failed_moves)
    env.close()
