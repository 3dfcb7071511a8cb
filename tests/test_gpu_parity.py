"""Parity tests proper: the HIP engine on a real MI355X, called through the C-ABI
(include/rware_hip.h via robotic-warehouse_amd/_capi.py), against
  (1) the golden vectors produced by the unmodified reference, and
  (2) the CPU oracle on the same seeded inputs at larger batches,
bit-exact on every field (grid, agent SoA, queue, counters, PCG64 state, obs, rewards, done)."""
import os
import re

import numpy as np
import pytest

import golden_util as gu
from engine_backend import EngineBackend
from rware_oracle import OracleVecEnv

import rware_amd

pytestmark = pytest.mark.gpu


def _loaded_native():
    from rware_amd import _capi
    return _capi.load()._name


@pytest.mark.parametrize("name", gu.fixture_names())
def test_engine_matches_reference_golden(name):
    meta, z = gu.load_fixture(name)
    be = EngineBackend(meta["E"], **gu.ctor_kwargs(meta))
    assert be.env.engines[0].info.arch_name.decode().startswith("gfx"), "not running on the HIP device"
    assert gu.replay(be, meta, z) == meta["T"]
    be.env.close()


@pytest.mark.parametrize("name,geom,tile", [
    ("small-4ag", (0, 0), 4), ("small-4ag", (0, 0), 2), ("small-4ag", (8, 256), 4), ("tiny-2ag", (0, 0), 4), ("medium-6ag-hard", (0, 0), 8),
    ("medium-6ag-hard", (16, 256), 16), ("large-16ag-sr2", (0, 0), 4),
    ("img-small-4ag-directional", (0, 0), 16), ("msg2-small-4ag", (0, 0), 16),
    # round 3: N = 8 in registers (ds_bpermute gathers, 64-bit chain links, row_half_mirror OR); Q > N (two queue slots per lane)
    ("small-8ag-global-inact", (0, 0), 16), ("tiny-4ag-easy-twostage", (0, 0), 16),
    # agent-count-static builds (N = 7 on 8-env workgroups; the large warehouse) and the 32-env 2-agent build
    ("small-7ag-hard", (0, 0), 4), ("large-4ag", (0, 0), 8), ("medium-2ag-easy", (32, 256), 8), ("medium-2ag-easy", (0, 0), 4),
    # round 4: every registered agent count in registers — N = 19 (128-bit chain links, three agent wavefronts) on both
    # geometries of its agent-count-static build; N = 16 exact build (BASELINE config 5) on its 4-env geometry
    ("small-19ag", (0, 0), 4), ("small-19ag", (4, 256), 2), ("large-16ag-sr2", (4, 256), 4),
])
def test_exact_shape_builds_match_reference_golden(name, geom, tile):
    """The golden traces of the unmodified reference on the EXACT-SHAPE kernel builds (what the BASELINE configs run):
    the fixture's few envs are tiled up to a whole number of workgroups; every field, every step."""
    meta, z = gu.load_fixture(name)
    be = EngineBackend(meta["E"], envs_per_workgroup=geom[0], threads_per_workgroup=geom[1], tile=tile, **gu.ctor_kwargs(meta))
    assert be.env.engines[0].info.specialised == 1
    assert gu.replay(be, meta, z) == meta["T"]
    be.env.close()


CASES = [
    # id, extra kwargs, B, T, geometry (envs/wg, threads/wg)
    ("rware-tiny-2ag-v1", {}, 4096, 560, (0, 0)),
    ("rware-small-4ag-v1", {}, 2048, 560, (0, 0)),
    ("rware-small-4ag-v1", {}, 1023, 120, (8, 64)),      # ragged batch: last workgroup is partial
    # exact-shape build with normalised (fractional) coordinates: the two-pass observation expansion
    ("rware-small-4ag-v1", {"normalised_coordinates": True}, 2048, 260, (0, 0)),
    ("rware-medium-6ag-hard-v1", {}, 1024, 300, (0, 0)),
    # the default geometry picks the half-size-workgroup builds at these batch sizes; pin the E = 16 builds too
    ("rware-small-4ag-v1", {}, 2048, 260, (16, 256)),
    ("rware-medium-6ag-hard-v1", {}, 1024, 200, (16, 256)),
    # further exact-shape builds (the tiny / small tasks of the RWARE benchmark suite)
    ("rware-tiny-4ag-v1", {}, 1024, 200, (0, 0)),
    ("rware-tiny-2ag-hard-v1", {}, 512, 200, (0, 0)),
    ("rware-tiny-4ag-hard-v1", {}, 512, 200, (0, 0)),
    ("rware-small-4ag-hard-v1", {}, 1024, 200, (0, 0)),
    # N > 32: one env per wavefront in the agent phases, 7-bit agent ids, crowded (chains and cycles every step)
    ("rware-medium-19ag-v1", {"n_agents": 64, "request_queue_size": 64, "max_steps": 70}, 256, 160, (0, 0)),
    ("rware-large-16ag-v1", {"sensor_range": 2}, 512, 200, (0, 0)),
    ("rware-small-19ag-v1", {"reward_type": 0, "max_inactivity_steps": 50}, 256, 200, (4, 128)),
    ("rware-tiny-4ag-easy-v1", {"reward_type": 2, "max_steps": 60}, 512, 200, (16, 256)),
    # > 255 shelves -> uint16 shelf shadow; 29 x 28 grid, 10 agents (the reference's __main__ smoke layout)
    ("rware-large-10ag-v1", {"shelf_columns": 9, "max_steps": 80, "sensor_range": 2}, 192, 200, (0, 0)),
]


@pytest.mark.parametrize("env_id,extra,B,T,geom", CASES)
@pytest.mark.parametrize("mode", ["next_step", "same_step"])
def test_engine_matches_oracle_large_batch(env_id, extra, B, T, geom, mode):
    kw = rware_amd.env_kwargs(env_id)
    kw.update(extra)
    kw["reward_type"] = rware_amd.enums.enum_value(kw["reward_type"])
    env = rware_amd.WarehouseVecEnv(B, autoreset_mode=mode, envs_per_workgroup=geom[0],
                                    threads_per_workgroup=geom[1], **kw)
    orc = OracleVecEnv(B, **kw)
    seed = 31337
    obs, _ = env.reset(seed=seed)
    assert np.array_equal(obs, orc.reset(seed=seed))
    rng = np.random.default_rng(5)
    N = kw["n_agents"]
    for t in range(T):
        p = [0.2] * 5 if (t // 50) % 2 == 0 else [0.1, 0.6, 0.1, 0.1, 0.1]
        a = rng.choice(5, size=(B, N), p=p).astype(np.int32)
        obs, rew, term, trunc, info = env.step(a)
        o2, r2, d2 = orc.step_autoreset(a, mode)
        assert np.array_equal(rew, r2), f"rewards t={t}"
        assert np.array_equal(term, d2.astype(bool)), f"done t={t}"
        assert np.array_equal(obs, o2), f"obs t={t}"
        if mode == "same_step":   # the pre-reset observation of the terminating step (rware/warehouse.py:929-946): info["final_obs"]
            assert ("final_obs" in info) == bool(d2.any()), t
            if d2.any():
                assert np.array_equal(info["_final_obs"], orc.final_mask)
                assert np.array_equal(info["final_obs"][orc.final_mask], orc.final_obs[orc.final_mask]), f"final_obs t={t}"
        if t % 25 == 0 or t == T - 1:
            st, so = env.get_state(), orc.get_state()
            for k in so:
                assert np.array_equal(st[k], so[k]), f"{k} t={t}"
    env.close()


def test_invalid_action_is_reported():
    env = rware_amd.make_vec("rware-tiny-2ag-v1", 8)
    env.reset(seed=0)
    with pytest.raises(ValueError):
        env.step(np.full((8, 2), 7))
    import torch
    tenv = rware_amd.make_vec("rware-tiny-2ag-v1", 8, output="torch")
    tenv.reset(seed=0)
    bad = torch.full((8, 2), 9, dtype=torch.int32, device="cuda")
    tenv.step(bad)
    with pytest.raises(ValueError):
        tenv.sync()
    tenv.close()
    env.close()


def test_torch_zero_copy_views_and_device_actions():
    import torch
    B = 256
    kw = rware_amd.env_kwargs("rware-small-4ag-v1")
    tenv = rware_amd.WarehouseVecEnv(B, output="torch", **kw)
    nenv = rware_amd.WarehouseVecEnv(B, **kw)
    o_t, _ = tenv.reset(seed=9)
    o_n, _ = nenv.reset(seed=9)
    assert o_t.is_cuda and o_t.shape == (B, 4, 71) and o_t.dtype == torch.float32
    assert np.array_equal(o_t.cpu().numpy(), o_n)
    g = torch.Generator(device="cpu").manual_seed(1)
    for _ in range(30):
        a = torch.randint(0, 5, (B, 4), generator=g, dtype=torch.int32)
        ot, rt, tt, _, _ = tenv.step(a.cuda() if _ % 2 else a.cuda().long())   # (int64, as a policy's argmax hands it over)
        on, rn, tn, _, _ = nenv.step(a.numpy())
        tenv.sync()
        assert np.array_equal(ot.cpu().numpy(), on) and np.array_equal(rt.cpu().numpy(), rn)
        assert np.array_equal(tt.cpu().numpy(), tn)
    assert ot.data_ptr() == tenv.device_tensor("obs").data_ptr()
    tenv.close(); nenv.close()


def test_torch_stream_ordering_without_sync():
    """output="torch": the engine enqueues on torch's CURRENT stream (RW_STREAM_USE_GIVEN; the default stream's handle
    is NULL), so a policy op producing the action tensor, the step, and a learner op consuming the observation tensor
    are ordered by the stream alone — no env.sync() anywhere.  A long chain of dependent steps would expose a race."""
    import torch
    B, T = 4096, 60
    kw = rware_amd.env_kwargs("rware-small-4ag-v1")
    tenv = rware_amd.WarehouseVecEnv(B, output="torch", **kw)
    nenv = rware_amd.WarehouseVecEnv(B, **kw)
    tenv.reset(seed=3)
    nenv.reset(seed=3)
    g = torch.Generator(device="cpu").manual_seed(2)
    tape = torch.randint(0, 5, (T, B, 4), generator=g, dtype=torch.int32)
    tape_dev = tape.cuda()
    torch.cuda.synchronize()
    sums, keep = [], []
    for t in range(T):
        # "policy": a torch op on the current stream writes the action tensor right before the step reads it
        a = (tape_dev[t] + torch.zeros((B, 4), dtype=torch.int32, device="cuda")).contiguous()
        keep.append(a)
        ot, rt, tt, _, _ = tenv.step(a)
        # "learner": torch ops on the same stream read the zero-copy views right after the step wrote them
        sums.append(torch.stack([ot.sum(dtype=torch.float64), rt.sum(dtype=torch.float64), tt.sum(dtype=torch.float64)]))
    got = torch.stack(sums).cpu().numpy()      # the only synchronisation: this copy
    for t in range(T):
        on, rn, tn, _, _ = nenv.step(tape[t].numpy())
        assert got[t, 0] == on.sum(dtype=np.float64) and got[t, 1] == rn.sum(dtype=np.float64) and got[t, 2] == tn.sum(), t
    # and on a non-default current stream
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        senv = rware_amd.WarehouseVecEnv(B, output="torch", **kw)
        senv.reset(seed=3)
        acc = []
        for t in range(20):
            a = (tape_dev[t] + 0).contiguous()
            keep.append(a)
            ot, rt, tt, _, _ = senv.step(a)
            acc.append(ot.sum(dtype=torch.float64))
        got2 = torch.stack(acc).cpu().numpy()
    assert np.array_equal(got2, got[:20, 0])
    tenv.close(); nenv.close(); senv.close()


def test_grid_view_is_refreshed_on_demand():
    """The exported int32 grid is a derived view (rw_refresh_grid): a zero-copy tensor obtained earlier lags behind later
    steps until env.refresh_grid(); get_state() and a fresh device_tensor("grid") are current."""
    import torch
    B = 1024
    kw = rware_amd.env_kwargs("rware-small-4ag-v1")
    env = rware_amd.WarehouseVecEnv(B, output="torch", **kw)
    orc = OracleVecEnv(B, **dict(kw, reward_type=kw["reward_type"].value))
    env.reset(seed=11)
    orc.reset(seed=11)
    g = env.device_tensor("grid")
    assert np.array_equal(g.cpu().numpy(), orc.get_state()["grid"])
    rng = np.random.default_rng(4)
    for t in range(20):
        a = rng.choice(5, size=(B, 4), p=[.1, .6, .1, .1, .1]).astype(np.int32)
        env.step(torch.from_numpy(a).cuda())
        orc.step_autoreset(a, "next_step")
    env.sync()
    want = orc.get_state()["grid"]
    assert not np.array_equal(g.cpu().numpy(), want)               # stale: the steps do not patch the view
    env.refresh_grid()
    assert np.array_equal(g.cpu().numpy(), want)
    assert np.array_equal(env.get_state()["grid"], want)
    env.close()


def test_full_size_headline_batch_properties():
    """BASELINE config 3 at full size (small-4ag, B=16384): size-independent invariants + an
    oracle spot-check of a strided subset of envs."""
    B, N = 16384, 4
    kw = rware_amd.env_kwargs("rware-small-4ag-v1")
    env = rware_amd.WarehouseVecEnv(B, **kw)
    env.reset(seed=0)
    rng = np.random.default_rng(12345)
    S = env.n_shelves
    acts = []
    for t in range(520):
        a = rng.integers(0, 5, size=(B, N), dtype=np.int32)
        acts.append(a)
        obs, rew, term, trunc, _ = env.step(a)
        assert term.all() == (t == 499) and term.any() == (t == 499)   # max_steps=500, all at once
    st = env.get_state()
    g = st["grid"]
    assert ((g[:, 0] > 0).sum(axis=(1, 2)) == N).all()                 # one cell per agent
    assert ((g[:, 1] > 0).sum(axis=(1, 2)) == S).all()                 # shelves are conserved
    ids = np.sort(g[:, 1].reshape(B, -1), axis=1)[:, -S:]
    assert (ids == np.arange(1, S + 1)).all()                          # each shelf id exactly once
    e = np.arange(B)[:, None]
    assert (g[e, 0, st["agent_y"], st["agent_x"]] == np.arange(1, N + 1)).all()
    carry = st["agent_carry"]
    assert ((carry == 0) | (g[e, 1, st["agent_y"], st["agent_x"]] == carry)).all()  # carried shelf under carrier
    q = np.sort(st["queue"], axis=1)
    assert (np.diff(q, axis=1) > 0).all() and q.min() >= 1 and q.max() <= S        # distinct requests
    assert (st["steps"] == 19).all()                                    # 520 steps = 500 + reset + 19
    sub = np.arange(0, B, 257)
    orc = OracleVecEnv(len(sub), **dict(kw, reward_type=kw["reward_type"].value))
    orc.seed(0)
    for i, e_ in enumerate(sub):
        from rware_oracle import seed_state
        orc.rng[i] = seed_state(int(e_))
    orc.reset()
    for a in acts:
        o2, _, _ = orc.step_autoreset(a[sub], "next_step")
    assert np.array_equal(obs[sub], o2)
    so = orc.get_state()
    for k in so:
        assert np.array_equal(st[k][sub], so[k]), k
    env.close()


@pytest.mark.parametrize("env_id,extra,B,mode", [
    ("rware-small-4ag-v1", {}, 4096, "next_step"),
    ("rware-tiny-2ag-v1", {"max_steps": 50}, 1000, "same_step"),
    ("rware-medium-6ag-hard-v1", {"reward_type": 0, "max_steps": 70}, 2048, "next_step"),
    ("rware-large-16ag-v1", {"sensor_range": 2}, 256, "next_step"),
])
def test_fused_rollout_matches_oracle(env_id, extra, B, mode):
    """rw_step_many_device: T steps in one launch with the env chunk resident in LDS == T oracle steps."""
    kw = rware_amd.env_kwargs(env_id)
    kw.update(extra)
    kw["reward_type"] = rware_amd.enums.enum_value(kw["reward_type"])
    env = rware_amd.WarehouseVecEnv(B, autoreset_mode=mode, **kw)
    orc = OracleVecEnv(B, **kw)
    assert np.array_equal(env.reset(seed=77)[0], orc.reset(seed=77))
    T = 530 if not extra.get("max_steps") else 160
    acts = np.random.default_rng(9).choice(5, size=(T, B, kw["n_agents"]), p=[.1, .55, .1, .1, .15]).astype(np.int32)
    obs, rew, term = env.rollout(acts)
    for t in range(T):
        o2, r2, d2 = orc.step_autoreset(acts[t], mode)
        assert np.array_equal(rew[t], r2) and np.array_equal(term[t], d2.astype(bool)), t
        assert np.array_equal(obs[t], o2), t
    st, so = env.get_state(), orc.get_state()
    for k in so:
        assert np.array_equal(st[k], so[k]), k
    env.close()


def test_snapshot_restore_on_device():
    B = 2048
    env = rware_amd.make_vec("rware-small-4ag-v1", B, max_steps=40)
    env.reset(seed=5)
    acts = np.random.default_rng(1).choice(5, size=(90, B, 4), p=[.1, .55, .1, .1, .15])
    for t in range(30):
        env.step(acts[t])
    snap, saved = env.snapshot(), env.get_state()
    first = env.rollout(acts[30:])
    env.restore(snap)
    back = env.get_state()
    for k in saved:
        assert np.array_equal(saved[k], back[k]), k
    second = env.rollout(acts[30:])
    for x, y in zip(first, second):
        assert np.array_equal(x, y)
    env.free_snapshot(snap)
    env.close()


@pytest.mark.parametrize("env_id,obs_type,directional,sr,B", [
    ("rware-small-4ag-v1", 2, True, 1, 4096),
    ("rware-medium-6ag-hard-v1", 3, False, 2, 1024),
])
def test_image_observations_match_oracle(env_id, obs_type, directional, sr, B):
    kw = rware_amd.env_kwargs(env_id)
    kw.update(sensor_range=sr, max_steps=60)
    kw["reward_type"] = kw["reward_type"].value
    extra = dict(observation_type=obs_type, image_observation_directional=directional)
    env = rware_amd.WarehouseVecEnv(B, **kw, **extra)
    orc = OracleVecEnv(B, **kw, **extra)

    def same(a, b):
        if isinstance(a, dict):
            return np.array_equal(a["image"], b[0]) and np.array_equal(a["features"], b[1])
        return np.array_equal(a, b)

    assert same(env.reset(seed=4)[0], orc.reset(seed=4))
    acts = np.random.default_rng(7).choice(5, size=(130, B, kw["n_agents"]), p=[.1, .5, .15, .15, .1])
    for t in range(130):
        o, r, d, _, _ = env.step(acts[t])
        o2, r2, d2 = orc.step_autoreset(acts[t], "next_step")
        assert same(o, o2) and np.array_equal(r, r2) and np.array_equal(d, d2.astype(bool)), t
    env.close()


SQUARE = dict(shelf_columns=3, column_height=3, shelf_rows=2, n_agents=5, msg_bits=0, sensor_range=2,
              request_queue_size=3, max_inactivity_steps=None, max_steps=60, reward_type=1)   # a 10 x 10 grid


@pytest.mark.parametrize("obs_type,directional,layers,B", [
    (2, True, [3, 4, 0, 2], 2048),      # AGENT_DIRECTION, AGENT_LOAD, SHELVES, AGENTS
    (3, False, [4, 5, 3], 1000),        # IMAGE_DICT, north-up, ragged batch
])
def test_transposed_image_layers_match_oracle(obs_type, directional, layers, B):
    """AGENT_DIRECTION / AGENT_LOAD exactly as the reference writes them (layer[ag.x, ag.y], :552/:558), on a
    square grid where that index is always in bounds; per-step and fused rollout."""
    extra = dict(observation_type=obs_type, image_observation_directional=directional, image_observation_layers=layers)
    env = rware_amd.WarehouseVecEnv(B, **SQUARE, **extra)
    orc = OracleVecEnv(B, **SQUARE, **extra)

    def same(a, b):
        if isinstance(a, dict):
            return np.array_equal(a["image"], b[0]) and np.array_equal(a["features"], b[1])
        return np.array_equal(a, b)

    assert same(env.reset(seed=4)[0], orc.reset(seed=4))
    acts = np.random.default_rng(7).choice(5, size=(150, B, 5), p=[.1, .45, .15, .15, .15])
    for t in range(110):
        o, r, d, _, _ = env.step(acts[t])
        o2, r2, d2 = orc.step_autoreset(acts[t], "next_step")
        assert same(o, o2) and np.array_equal(r, r2) and np.array_equal(d, d2.astype(bool)), t
    img, rew, term = env.rollout(acts[110:])
    for t in range(110, 150):
        o2, r2, d2 = orc.step_autoreset(acts[t], "next_step")
        assert np.array_equal(img[t - 110], o2[0] if isinstance(o2, tuple) else o2) and np.array_equal(rew[t - 110], r2), t
    env.close()


@pytest.mark.parametrize("layer", [3, 4])
def test_transposed_image_layers_raise_indexerror_like_the_reference(layer):
    """Registered layouts have H > W: the reference's transposed write raises IndexError once an agent (a loaded
    one for AGENT_LOAD) stands at y >= W; the engine reports it for the same reset()/step() call."""
    kw = rware_amd.env_kwargs("rware-small-4ag-v1")
    kw["reward_type"] = kw["reward_type"].value
    extra = dict(observation_type=2, image_observation_layers=[2, layer])
    B = 64 if layer == 3 else 8
    env = rware_amd.WarehouseVecEnv(B, **kw, **extra)
    orc = OracleVecEnv(B, **kw, **extra)
    rng = np.random.default_rng(1)

    def attempt(f):
        try:
            return f(), False
        except IndexError:
            return None, True

    (o, e1), (o2, e2) = attempt(lambda: env.reset(seed=2)[0]), attempt(lambda: orc.reset(seed=2))
    assert e1 == e2
    t = 0
    while not e1 and t < 600:
        a = rng.choice(5, size=(B, 4), p=[.05, .5, .15, .15, .15])
        (res, e1), (res2, e2) = attempt(lambda: env.step(a)), attempt(lambda: orc.step_autoreset(a, "next_step"))
        assert e1 == e2, t
        if not e1:
            assert np.array_equal(res[0], res2[0]), t
        t += 1
    assert e1, "no agent ever reached y >= W"
    env.close()


def test_dict_observations_are_the_unflattened_oracle_vector():
    """ObservationType.DICT: host views of the FLATTENED batch (whose equality with the reference's flatten(DICT)
    the golden replay establishes; the dict itself is checked against the live reference in the CPU suite)."""
    kw = rware_amd.env_kwargs("rware-small-4ag-v1")
    kw["reward_type"] = kw["reward_type"].value
    B = 512
    env = rware_amd.WarehouseVecEnv(B, observation_type=rware_amd.ObservationType.DICT, **kw)
    orc = OracleVecEnv(B, **kw)
    d, _ = env.reset(seed=9)
    flat = orc.reset(seed=9)
    rng = np.random.default_rng(2)
    for t in range(40):
        if t:
            a = rng.integers(0, 5, size=(B, 4))
            d = env.step(a)[0]
            flat = orc.step_autoreset(a, "next_step")[0]
        assert np.array_equal(d["self"]["location"], flat[..., :2].astype(np.int32))
        assert np.array_equal(d["self"]["direction"], flat[..., 3:7].argmax(-1))
        assert np.array_equal(d["self"]["carrying_shelf"][..., 0], flat[..., 2]) and np.array_equal(d["self"]["on_highway"][..., 0], flat[..., 7])
        for c, cell in enumerate(d["sensors"]):
            o = 8 + 7 * c
            assert np.array_equal(cell["has_agent"][..., 0], flat[..., o]) and np.array_equal(cell["direction"], flat[..., o + 1:o + 5].argmax(-1))
            assert np.array_equal(cell["has_shelf"][..., 0], flat[..., o + 5]) and np.array_equal(cell["shelf_requested"][..., 0], flat[..., o + 6])
            assert cell["local_message"] is None
    env.close()


@pytest.mark.parametrize("env_id,extra,B,T", [
    ("rware-small-4ag-v1", {}, 16384, 1100),                          # BASELINE config 3 (headline): two mass autoresets
    ("rware-tiny-2ag-v1", {}, 4096, 560),                             # config 2
    ("rware-medium-6ag-hard-v1", {}, 8192, 560),                      # config 4, per-GPU shard of 65536 / 8
    ("rware-large-16ag-v1", {"sensor_range": 2}, 16384, 520),         # config 5, per-GPU shard of 131072 / 8
])
def test_full_batch_soak_every_env_against_oracle(env_id, extra, B, T):
    """Every BASELINE config at its full per-GPU batch: EVERY env's rewards, done flags AND observations each step, and the
    complete final state, against the oracle; per-step launches interleaved with fused 40-step rollouts (the first of them with
    its observation tape, every step of it compared), across the mass autoreset at step 500."""
    kw = rware_amd.env_kwargs(env_id)
    kw.update(extra)
    N = kw["n_agents"]
    env = rware_amd.WarehouseVecEnv(B, **kw)
    assert env.engines[0].info.specialised == 1
    orc = OracleVecEnv(B, **dict(kw, reward_type=kw["reward_type"].value))
    assert np.array_equal(env.reset(seed=2024)[0], orc.reset(seed=2024))
    rng = np.random.default_rng(99)
    t = 0
    while t < T:
        if t % 100 < 60:   # stepwise launches
            a = rng.choice(5, size=(B, N), p=[.1, .5, .15, .15, .1]).astype(np.int32)
            obs, rew, term, _, _ = env.step(a)
            o2, r2, d2 = orc.step_autoreset(a, "next_step")
            assert np.array_equal(rew, r2) and np.array_equal(term, d2.astype(bool)), t
            assert np.array_equal(obs, o2), t          # (every step: VERDICT r4 item 3 — it was every 25th)
            t += 1
        else:              # a fused 40-step rollout; the first one hands its observation tape back as well
            acts = rng.choice(5, size=(40, B, N), p=[.1, .5, .15, .15, .1]).astype(np.int32)
            with_obs = t < 100
            n_obs = 40 if B * N * env.engines[0].L * 4 * 40 < 2e9 else 8   # (config 5: 192 MB per step — the first 8 steps' worth)
            if with_obs and n_obs < 40:   # two launches: 8 steps with observations, 32 without
                otape, rew_a, term_a = env.rollout(acts[:n_obs], want_obs=True)
                _, rew_b, term_b = env.rollout(acts[n_obs:], want_obs=False)
                rew, term = np.concatenate([rew_a, rew_b]), np.concatenate([term_a, term_b])
            else:
                otape, rew, term = env.rollout(acts, want_obs=with_obs)
            for k in range(40):
                o2, r2, d2 = orc.step_autoreset(acts[k], "next_step")
                assert np.array_equal(rew[k], r2) and np.array_equal(term[k], d2.astype(bool)), t + k
                if with_obs and k < n_obs:
                    assert np.array_equal(otape[k], o2), t + k
            t += 40
    st, so = env.get_state(), orc.get_state()
    for k in so:
        assert np.array_equal(st[k], so[k]), k
    assert np.array_equal(env.observations(), orc.obs())
    env.close()


def test_hip_graph_capture_of_per_step_launches():
    """rw_step_device only enqueues a kernel, so a training loop can capture env steps (with its policy) in a HIP graph:
    100 steps captured once and replayed == the same steps launched one by one."""
    import torch
    B, K = 2048, 100
    kw = rware_amd.env_kwargs("rware-small-4ag-v1")
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        genv = rware_amd.WarehouseVecEnv(B, output="torch", **kw)
        penv = rware_amd.WarehouseVecEnv(B, output="torch", **kw)
        genv.reset(seed=21)
        penv.reset(seed=21)
        tape = torch.randint(0, 5, (K, B, 4), dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            genv.engines[0].step_tape_device(tape.data_ptr(), K, 0, K)
        for rep in range(3):   # 300 steps, across a mass autoreset... max_steps = 500: no; replay thrice anyway
            g.replay()
            for t in range(K):
                penv.engines[0].step_device(tape[t].data_ptr())
        torch.cuda.synchronize()
        for name in ("obs", "rewards", "terminated"):
            assert torch.equal(genv.device_tensor(name), penv.device_tensor(name)), name
    a, b = genv.get_state(), penv.get_state()
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    genv.close(); penv.close()


def test_timed_tape_launches_carry_their_own_events():
    """rw_step_tape_device_timed: the same launches as rw_step_tape_device; the start / stop events ride on the first /
    last dispatch, so their elapsed time is the device time of the n launches (a few us each) with no marker packets."""
    import torch
    B, K = 4096, 40
    kw = rware_amd.env_kwargs("rware-small-4ag-v1")
    a = rware_amd.WarehouseVecEnv(B, **kw)
    b = rware_amd.WarehouseVecEnv(B, **kw)
    a.reset(seed=5)
    b.reset(seed=5)
    tape = torch.randint(0, 5, (K, B, 4), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    ea, eb = a.engines[0], b.engines[0]
    ea.step_tape_device_timed(tape.data_ptr(), K, 0, K, 0, 1)
    eb.step_tape_device(tape.data_ptr(), K, 0, K)
    ea.sync(); eb.sync()
    ms = ea.event_elapsed_ms(0, 1)
    assert 0.0 < ms < 50.0 and ms / K > 0.001       # 40 launches of >= 1 us each, well under 50 ms
    ea.step_tape_device_timed(tape.data_ptr(), K, 0, 1, 2, 3)    # one launch carries both events
    eb.step_tape_device(tape.data_ptr(), K, 0, 1)
    ea.sync(); eb.sync()
    assert 0.0 < ea.event_elapsed_ms(2, 3) < 5.0
    assert np.array_equal(a.observations(), b.observations())
    sa, sb = a.get_state(), b.get_state()
    for k in sa:
        assert np.array_equal(sa[k], sb[k]), k
    a.close(); b.close()


def _no_frac_above_one(d, path=""):
    """every key named frac* in the line is a fraction of a bandwidth — only frac_algorithmic (SURVEY.md §8(d)'s
    bytes, which the engine does not move: a work rate) may exceed 1."""
    if isinstance(d, dict):
        for k, v in d.items():
            if "frac" in k and "algorithmic" not in k and isinstance(v, (int, float)):
                assert 0 <= v <= 1.0, f"{path}.{k} = {v}"
            _no_frac_above_one(v, f"{path}.{k}")
    elif isinstance(d, list):
        for i, v in enumerate(d):
            _no_frac_above_one(v, f"{path}[{i}]")


def _check_bench_line(out, n_gpus, steps, warmup, batch):
    import json
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]          # rank 0 prints ONE line
    d = json.loads(lines[0])
    assert d["n_gpus"] == n_gpus and d["steps"] == steps and d["warmup"] == warmup and d["scaling"] == "weak"
    assert d["unit"] == "agent-steps/s" and d["higher_is_better"] is True and "cpu_baseline" not in d
    expect = n_gpus * batch * 4 * steps / (d["ms_per_step"] * 1e-3 * steps)
    assert abs(d["value"] - expect) / expect < 1e-6     # whole-job aggregate over all ranks
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["frac"] > 0
    _no_frac_above_one(d)
    assert d["roofline"]["regime"] in ("infinity-cache", "hbm") and d["roofline"]["basis"] in ("pmc-traffic", "engine-bytes")
    # the practical floor beside the 8 TB/s one: a kernel that only writes this step's observations, same geometry, launched the same way
    assert 0 < d["roofline"]["store_only_kernel_ms_per_launch"] < d["roofline"]["kernel_ms_per_launch"]
    assert "native loop" in d["config"]["submit"]        # the SAME submission path at N = 1 and N > 1 ...
    for r in d["ranks"]:                                 # ... and the graph path + the host's cost per launch beside it, per rank
        assert r["host_issue_us_per_step"] > 0 and r["graph_ms_per_step"] > 0 and r["graph_launches_per_replay"] == min(steps, 256), r
    # priced on the bytes the engine's layout has to move: a fraction of a bandwidth, never above 1 (a timed kernel that
    # skipped work, or a byte model that over-counts, would show here)
    assert 0 < d["roofline"]["frac_engine"] <= 1.0 and 0 < d["sustained"]["roofline_frac_engine"] <= 1.0
    assert d["roofline"]["engine_bytes_per_launch"] < d["roofline"]["algorithmic_bytes_per_launch"]
    assert d["roofline"]["peak_measured"] == 6290.0 and "frac_physical" in d["roofline"]
    sus = d["sustained"]                                 # the fixed-length steady-state leg rides in the same line
    assert sus["steps"] == 2000 and sus["warmup"] == 100 and sus["ms_per_step"] > 0 and sus["kernel_ms_per_launch"] > 0
    assert "RCCL" not in d["config"]["parallelism"] or "no RCCL" in d["config"]["parallelism"]
    return d


def test_bench_two_ranks_self_spawned():
    """`python bench.py --gpus 2` with NO launcher (how the driver starts N = 1): bench.py spawns the second rank
    itself; barrier + MAX over gloo.  Both ranks share this box's single device (test hook)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["RWARE_BENCH_SHARE_GPU"] = "1"
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "200", "--warmup", "20", "--batch", "4096", "--no-soak"]
    out = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=600)
    _check_bench_line(out, 2, 200, 20, 4096)


def test_bench_graph_submit_mode():
    """`bench.py --submit graph` (the per-step launches replayed from a HIP graph — what profiles/tools/sweep.sh traces
    the small kernels with): a valid line, and the same step time as the native loop within a wide margin."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["RWARE_BENCH_TAPE_STEPS"] = "64"
    res = {}
    for mode in ("graph", "native"):
        cmd = [sys.executable, os.path.join(root, "bench.py"), "--submit", mode, "--steps", "700", "--warmup", "30", "--batch", "4096",
               "--no-cpu-baseline", "--no-hbm-regime", "--no-api-loop", "--no-fused-extra", "--no-sustained", "--no-soak"]
        out = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        res[mode] = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
        assert res[mode]["steps"] == 700 and res[mode]["ms_per_step"] > 0
    assert "HIP graph" in res["graph"]["config"]["submit"]
    assert 0.5 < res["graph"]["ms_per_step"] / res["native"]["ms_per_step"] < 2.0


def test_bench_two_ranks_through_torchrun():
    """The driver's N > 1 invocation of bench.py (torch.distributed.run, one rank per GPU), exercised on this
    1-GPU box: both ranks share the device (test hook).  Checks the contract of the JSON line and that rank 0
    reports the whole-job rate."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RWARE_BENCH_SHARE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29731", os.path.join(root, "bench.py"),
           "--gpus", "2", "--steps", "200", "--warmup", "20", "--batch", "4096", "--no-soak"]
    out = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=600)
    _check_bench_line(out, 2, 200, 20, 4096)


# --------------------------------------------------------------------------- round 3: the Python surface a training loop calls
def test_step_calls_enqueue_exactly_one_kernel_each():
    """100 `WarehouseVecEnv(output="torch").step(cuda int32 actions)` calls put exactly 100 kernels on the stream: the step
    kernel and nothing else (the flags come back as zero-copy bool views, not through cast kernels)."""
    import torch
    from torch.profiler import ProfilerActivity, profile
    B = 4096
    kw = rware_amd.env_kwargs("rware-small-4ag-v1")
    env = rware_amd.WarehouseVecEnv(B, output="torch", **kw)
    env.reset(seed=1)
    acts = torch.randint(0, 5, (8, B, 4), dtype=torch.int32, device="cuda")
    slices = [acts[t] for t in range(8)]
    for t in range(8):
        env.step(slices[t])
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        for t in range(100):
            obs, rew, term, trunc, _ = env.step(slices[t % 8])
        torch.cuda.synchronize()
    assert term.dtype == torch.bool and trunc.dtype == torch.bool and not bool(trunc.any())
    assert term.data_ptr() == env.device_tensor("terminated").data_ptr()      # a view of the engine's buffer, not a copy
    evs = prof.events()
    kernels = [e for e in evs if str(getattr(e, "device_type", "")).endswith("CUDA") and "memcpy" not in e.name.lower()
               and "memset" not in e.name.lower()]
    # torch ops that would mean a kernel or a copy per step on the host side of the trace
    bad_ops = [e.name for e in evs if e.name in ("aten::to", "aten::_to_copy", "aten::copy_", "aten::contiguous", "aten::clone")]
    assert not bad_ops, bad_ops[:5]
    if kernels:  # (GPU activity records need the ROCm profiler library behind kineto; the host-op check above always runs)
        names = {e.name for e in kernels}
        assert len(kernels) == 100, (len(kernels), sorted(names)[:5])
        assert all("rware_step_kernel" in n for n in names), names
    env.close()


def test_device_rollout_takes_and_returns_torch_tensors():
    """rollout() with a CUDA action tape: torch tapes out (no host copy, no sync inside), bit-equal to the host-array rollout
    and to T single steps; the host path's device tapes come from a grow-only arena (no allocation after the first call)."""
    import torch
    B, T = 2048, 48
    kw = rware_amd.env_kwargs("rware-small-4ag-v1")
    kw["max_steps"] = 40                                  # crosses an autoreset inside the fused launch
    tenv = rware_amd.WarehouseVecEnv(B, output="torch", **kw)
    nenv = rware_amd.WarehouseVecEnv(B, **kw)
    senv = rware_amd.WarehouseVecEnv(B, **kw)
    for e in (tenv, nenv, senv):
        e.reset(seed=5)
    acts = np.random.default_rng(3).choice(5, size=(2 * T, B, 4), p=[.1, .55, .1, .1, .15]).astype(np.int32)
    for half in range(2):
        a = acts[half * T:(half + 1) * T]
        ot, rt, tt = tenv.rollout(torch.from_numpy(a).cuda())
        assert ot.is_cuda and ot.shape == (T, B, 4, 71) and rt.shape == (T, B, 4) and tt.dtype == torch.bool and tt.shape == (T, B)
        on, rn, tn = nenv.rollout(a)
        assert np.array_equal(ot.cpu().numpy(), on) and np.array_equal(rt.cpu().numpy(), rn) and np.array_equal(tt.cpu().numpy(), tn)
        for t in range(T):
            o1, r1, d1, _, _ = senv.step(a[t])
            assert np.array_equal(on[t], o1) and np.array_equal(rn[t], r1) and np.array_equal(tn[t], d1), (half, t)
    assert nenv.engines[0].arena_allocations == 4          # actions, obs, rewards, terminated: once, not once per call
    _, r64, _ = tenv.rollout(torch.from_numpy(acts[:T]).cuda().long(), want_obs=False)   # int64 tape, no observations
    assert r64.shape == (T, B, 4)
    tenv.sync()
    tenv.close(); nenv.close(); senv.close()


def test_multi_device_torch_output_launches_every_shard():
    """One process driving several devices with device-resident results: step() takes one CUDA tensor per device, returns one
    tensor per device, and every launch is enqueued before anything waits.  This box has one GPU: both shards sit on device 0
    (two engines, two buffers); the concatenation equals the unsharded env."""
    import torch
    B = 1024
    kw = rware_amd.env_kwargs("rware-small-4ag-v1")
    menv = rware_amd.WarehouseVecEnv(B, devices=[0, 0], output="torch", **kw)
    nenv = rware_amd.WarehouseVecEnv(B, **kw)
    assert menv.shard_bounds == [(0, 512), (512, 1024)]
    om, _ = menv.reset(seed=11)
    on, _ = nenv.reset(seed=11)
    assert isinstance(om, tuple) and len(om) == 2 and om[0].shape == (512, 4, 71)
    assert np.array_equal(torch.cat(om).cpu().numpy(), on)
    rng = np.random.default_rng(0)
    for t in range(40):
        a = rng.integers(0, 5, size=(B, 4), dtype=np.int32)
        parts = [torch.from_numpy(a[lo:hi]).cuda() for lo, hi in menv.shard_bounds]
        om, rm, tm, um, _ = menv.step(parts if t % 2 else a)      # per-device tensors, or one host array
        on, rn, tn, _, _ = nenv.step(a)
        assert np.array_equal(torch.cat(om).cpu().numpy(), on) and np.array_equal(torch.cat(rm).cpu().numpy(), rn), t
        assert np.array_equal(torch.cat(tm).cpu().numpy(), tn) and tm[0].dtype == torch.bool
    with pytest.raises(ValueError):
        menv.step(torch.zeros((B, 4), dtype=torch.int32, device="cuda"))   # one tensor for two devices
    # numpy output across shards: the per-device read-backs run concurrently (one reader thread per device)
    henv = rware_amd.WarehouseVecEnv(B, devices=[0, 0], **kw)
    oh, _ = henv.reset(seed=11)
    n2 = rware_amd.WarehouseVecEnv(B, **kw)
    o2, _ = n2.reset(seed=11)
    assert np.array_equal(oh, o2)
    for t in range(10):
        a = rng.integers(0, 5, size=(B, 4), dtype=np.int32)
        oh, rh, th, _, _ = henv.step(a)
        o2, r2, t2, _, _ = n2.step(a)
        assert np.array_equal(oh, o2) and np.array_equal(rh, r2) and np.array_equal(th, t2)
    menv.close(); nenv.close(); henv.close(); n2.close()


def test_bench_eight_ranks_report_placement_and_per_rank_times():
    """`bench.py --gpus 8` (self-spawned; all ranks on this box's one device): the same native submission path as N = 1, every rank
    pins itself to physical cores of its GPU's NUMA node, and the line lists every rank's own time, CPUs, host cost per launch and
    the HIP-graph replay time of the same launches."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["RWARE_BENCH_SHARE_GPU"] = "1"
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "300", "--warmup", "20", "--batch", "2048", "--no-soak"]
    out = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=900)
    d = _check_bench_line(out, 8, 300, 20, 2048)
    ranks = d["ranks"]
    assert [r["rank"] for r in ranks] == list(range(8))
    assert all(r["ms_per_step"] > 0 and r["kernel_ms_per_launch"] > 0 for r in ranks)
    assert max(r["ms_per_step"] for r in ranks) <= d["ms_per_step"] * (1 + 1e-9)          # `value` is the MAX over ranks
    pinned = [r["placement"] for r in ranks if r["placement"].get("pinned")]
    if len(pinned) == 8:  # (a box with fewer than 8 physical cores reports why it did not pin)
        masks = [p["cpus"] if isinstance(p["cpus"], list) else p["cpus"] for p in pinned]
        assert len({str(m) for m in masks}) == 8, masks                                  # disjoint core sets


def test_bench_driver_invocation_carries_every_leg():
    """The driver's own N = 1 call, `bench.py --steps 20 --warmup 5`: ONE line with roofline, sustained, the HBM-regime leg,
    the closed-loop API leg, and the CPU baseline north_star names — the reference's pure-Python step (staged, unmodified,
    under the git-ignored oracle/_ref) timed on THIS box's cores, the C port beside it."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "20", "--warmup", "5"], env=env, cwd=root,
                         capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5 and d["config"]["envs_per_gpu"] == 16384
    h = d["hbm_regime"]
    assert h["steps"] >= 200 and 0.02 < h["kernel_ms_per_launch"] < 0.5 and 0.2 < h["frac"] <= 1.0 and h["frac_algorithmic"] > h["frac"]
    assert h["regime"] == "hbm" and d["roofline"]["regime"] == "infinity-cache"   # 376 MB per step vs 23 MB
    _no_frac_above_one(d)
    assert d["ranks"][0]["graph_launches_per_replay"] == 20 and d["ranks"][0]["graph_ms_per_step"] > 0   # the 20-step graph replays
    k = d["soak"]   # >= 3 s of back-to-back launches: an outside utilisation sampler can see the GPU busy; steps x ms = wall
    assert k["wall_s"] >= 3.0 and abs(k["steps"] * k["ms_per_step"] * 1e-3 - k["wall_s"]) < 1e-6 * k["wall_s"] + 1e-9
    assert 0.5 < k["kernel_ms_per_launch"] / d["sustained"]["kernel_ms_per_launch"] < 1.5
    a = d["api_closed_loop"]
    assert a["steps"] == 2000 and 3.0 < a["us_per_step"] < 100.0
    c = d["cpu_baseline"]
    assert c["cores"] >= 1 and c["port"]["aggregate"] > 0
    if os.path.isfile(os.path.join(root, "oracle", "_ref", "rware", "warehouse.py")):
        assert c["kind"] == "reference" and c["value"] > 0 and "oracle/_ref" in c["reference_root"], c


PAPER_GRID = [f"rware-{size}-{n}ag{diff}-v1" for size in ("tiny", "small", "medium") for n in (2, 4, 6, 8) for diff in ("-easy", "", "-hard")]


@pytest.mark.parametrize("env_id", PAPER_GRID)
def test_paper_task_grid_exact_shape_builds_match_oracle(env_id):
    """Every task of the RWARE papers' grid (tiny / small / medium x 2, 4, 6, 8 agents x easy / normal / hard) runs an exact-shape
    kernel build; each against the oracle on every env, per-step launches and a fused rollout, across autoresets."""
    kw = rware_amd.env_kwargs(env_id)
    kw["max_steps"] = 50
    kw["reward_type"] = rware_amd.enums.enum_value(kw["reward_type"])
    B, N = 512, kw["n_agents"]
    # 6 and 8 agents have two builds (8 envs per workgroup up to 16384 envs, 16 beyond): alternate which one is pinned
    # (2 agents: 16 envs per workgroup below 16384 envs, 32 from there on — every 2-agent id is pinned to the 32-env build here)
    geom = (16, 256) if N >= 6 and len(env_id) % 2 else (32, 256) if N == 2 else (0, 0)
    env = rware_amd.WarehouseVecEnv(B, envs_per_workgroup=geom[0], threads_per_workgroup=geom[1], **kw)
    assert env.engines[0].info.specialised == 1
    assert env.engines[0].info.envs_per_workgroup == (geom[0] if geom[0] else 16 if N < 6 else 8)
    orc = OracleVecEnv(B, **kw)
    assert np.array_equal(env.reset(seed=31)[0], orc.reset(seed=31))
    rng = np.random.default_rng(8)
    for t in range(70):
        a = rng.choice(5, size=(B, N), p=[.1, .55, .1, .1, .15]).astype(np.int32)
        obs, rew, term, _, _ = env.step(a)
        o2, r2, d2 = orc.step_autoreset(a, "next_step")
        assert np.array_equal(obs, o2) and np.array_equal(rew, r2) and np.array_equal(term, d2.astype(bool)), t
    acts = rng.choice(5, size=(40, B, N), p=[.1, .55, .1, .1, .15]).astype(np.int32)
    obs, rew, term = env.rollout(acts)
    for k in range(40):
        o2, r2, d2 = orc.step_autoreset(acts[k], "next_step")
        assert np.array_equal(obs[k], o2) and np.array_equal(rew[k], r2) and np.array_equal(term[k], d2.astype(bool)), k
    st, so = env.get_state(), orc.get_state()
    for k in so:
        assert np.array_equal(st[k], so[k]), k
    env.close()


OFF_GRID = ["rware-tiny-3ag-v1", "rware-tiny-7ag-easy-v1", "rware-small-1ag-hard-v1", "rware-small-5ag-v1", "rware-medium-3ag-hard-v1",
            "rware-medium-7ag-v1", "rware-large-2ag-v1", "rware-large-4ag-easy-v1", "rware-large-6ag-hard-v1", "rware-large-8ag-v1",
            # round 4: 9 .. 19 agents, every count once, all four sizes, the three difficulties
            "rware-small-9ag-hard-v1", "rware-small-10ag-v1", "rware-tiny-11ag-v1", "rware-small-12ag-easy-v1", "rware-medium-13ag-v1",
            "rware-large-14ag-hard-v1", "rware-tiny-15ag-easy-v1", "rware-large-16ag-v1", "rware-medium-17ag-easy-v1",
            "rware-small-18ag-v1", "rware-small-19ag-v1", "rware-large-19ag-hard-v1"]


@pytest.mark.parametrize("env_id", OFF_GRID)
def test_agent_count_static_builds_match_oracle(env_id):
    """Registered ids without an exact (N, Q) entry — odd agent counts, the large warehouse — run the agent-count-static
    build of their size and agent count (request-queue length read at run time): against the oracle on every env, per-step
    launches and a fused rollout, across autoresets."""
    kw = rware_amd.env_kwargs(env_id)
    kw["max_steps"] = 45
    kw["reward_type"] = rware_amd.enums.enum_value(kw["reward_type"])
    B, N = 512, kw["n_agents"]
    env = rware_amd.WarehouseVecEnv(B, **kw)
    assert env.engines[0].info.build_kind == 2
    orc = OracleVecEnv(B, **kw)
    assert np.array_equal(env.reset(seed=41)[0], orc.reset(seed=41))
    rng = np.random.default_rng(9)
    for t in range(60):
        a = rng.choice(5, size=(B, N), p=[.1, .5, .1, .1, .2]).astype(np.int32)
        obs, rew, term, _, _ = env.step(a)
        o2, r2, d2 = orc.step_autoreset(a, "next_step")
        assert np.array_equal(obs, o2) and np.array_equal(rew, r2) and np.array_equal(term, d2.astype(bool)), t
    acts = rng.choice(5, size=(40, B, N), p=[.1, .5, .1, .1, .2]).astype(np.int32)
    obs, rew, term = env.rollout(acts)
    for k in range(40):
        o2, r2, d2 = orc.step_autoreset(acts[k], "next_step")
        assert np.array_equal(obs[k], o2) and np.array_equal(rew[k], r2) and np.array_equal(term[k], d2.astype(bool)), k
    st, so = env.get_state(), orc.get_state()
    for k in so:
        assert np.array_equal(st[k], so[k]), k
    env.close()


@pytest.mark.parametrize("env_id,p_forward,geom", [
    ("rware-tiny-9ag-v1", 0.8, (0, 0)), ("rware-tiny-12ag-easy-v1", 0.75, (0, 0)), ("rware-tiny-13ag-v1", 0.75, (0, 0)),
    ("rware-tiny-16ag-v1", 0.75, (0, 0)), ("rware-tiny-17ag-hard-v1", 0.7, (0, 0)), ("rware-tiny-19ag-v1", 0.8, (0, 0)),
    ("rware-small-16ag-v1", 0.8, (4, 256)), ("rware-small-19ag-v1", 0.8, (4, 256)),
])
def test_crowded_warehouses_resolve_long_chains(env_id, p_forward, geom):
    """9 .. 19 agents on the 110 cells of the tiny warehouse (and the 4-env geometry on the small one) under a forward-heavy
    policy: long follower chains, contested cells with unequal depths, blocked tails and cycles on nearly every step — the
    per-cell agent phases of the per-step kernels and the register-exchange ones of the fused rollout (wide priority words,
    64- / 128-bit chain links) against the oracle's literal networkx restatement, every env, every step."""
    kw = rware_amd.env_kwargs(env_id)
    kw["max_steps"] = 60
    kw["reward_type"] = rware_amd.enums.enum_value(kw["reward_type"])
    B, N = 256, kw["n_agents"]
    env = rware_amd.WarehouseVecEnv(B, envs_per_workgroup=geom[0], threads_per_workgroup=geom[1], **kw)
    assert env.engines[0].info.build_kind == 2 and env.engines[0].info.envs_per_workgroup == (geom[0] or 8)
    orc = OracleVecEnv(B, **kw)
    assert np.array_equal(env.reset(seed=31)[0], orc.reset(seed=31))
    rng = np.random.default_rng(33)
    rest = (1.0 - p_forward) / 4
    for t in range(150):
        a = rng.choice(5, size=(B, N), p=[rest, p_forward, rest, rest, rest]).astype(np.int32)
        obs, rew, term, _, _ = env.step(a)
        o2, r2, d2 = orc.step_autoreset(a, "next_step")
        assert np.array_equal(obs, o2) and np.array_equal(rew, r2) and np.array_equal(term, d2.astype(bool)), t
    acts = rng.choice(5, size=(25, B, N), p=[rest, p_forward, rest, rest, rest]).astype(np.int32)
    obs, rew, term = env.rollout(acts)                      # the fused rollout keeps the all-gather (register) agent phases
    for k in range(25):
        o2, r2, d2 = orc.step_autoreset(acts[k], "next_step")
        assert np.array_equal(obs[k], o2) and np.array_equal(rew[k], r2) and np.array_equal(term[k], d2.astype(bool)), k
    st, so = env.get_state(), orc.get_state()
    for k in so:
        assert np.array_equal(st[k], so[k]), k
    env.close()


def test_dpp_subrev_probe_on_this_gpu(tmp_path):
    """The compiler / hardware finding the agent phases are written around (rware_kernels.h, P2b): a DPP cross-lane move folded
    into a subtract (`v_subrev_u32_dpp`) comes out with its operands swapped on gfx950.  The probe is built and run HERE, on
    the GPU: the form the kernels use (xor-compare on gathered values that sit in registers of their own) must be exact, a
    plain `lane0 - me` as well; what the folded subtract does on this toolchain is recorded in the message."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this box")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "dpp_subrev_probe")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-o", exe, os.path.join(root, "profiles", "tools", "dpp_subrev_probe.hip")],
                   check=True, capture_output=True, timeout=600)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120).stdout
    m = re.search(r"folded in (\d+) / 64, xor-compare on unfolded values (\d+) / 64, plain lane0 - me (\d+) / 64", out)
    assert m, out
    assert int(m.group(2)) == 0 and int(m.group(3)) == 0, out       # what the kernels rely on
    print("dpp_subrev_probe:", out.strip())                          # (32 / 64 wrong lanes for the folded form on ROCm 7.2)


@pytest.mark.parametrize("env_id,extra,B", [("rware-small-4ag-v1", {}, 4096), ("rware-small-4ag-v1", {"observation_type": 2}, 2048),
                                            ("rware-large-16ag-v1", {"sensor_range": 2}, 512), ("rware-small-12ag-v1", {}, 1024)])
def test_observation_store_policy_does_not_change_results(env_id, extra, B):
    """Non-temporal vs cached observation stores (two builds of the exact kernels, a run-time switch in the others): the
    default rule, forced `cached` and forced `stream` give bit-identical steps against the oracle."""
    kw = rware_amd.env_kwargs(env_id)
    kw.update(extra)
    kw["max_steps"] = 30
    N = kw["n_agents"]
    envs = {pol: rware_amd.WarehouseVecEnv(B, obs_stores=pol, **kw) for pol in (None, "cached", "stream")}
    assert envs["cached"].engines[0].info.obs_stores_stream == 0 and envs["stream"].engines[0].info.obs_stores_stream == 1
    okw = {k: v for k, v in kw.items() if k != "observation_type"}
    okw["reward_type"] = rware_amd.enums.enum_value(okw["reward_type"])
    orc = OracleVecEnv(B, **okw) if "observation_type" not in extra else None
    obs = {pol: e.reset(seed=6)[0] for pol, e in envs.items()}
    if orc is not None:
        assert np.array_equal(obs[None], orc.reset(seed=6))
    rng = np.random.default_rng(2)
    for t in range(45):
        a = rng.choice(5, size=(B, N), p=[.1, .5, .1, .1, .2]).astype(np.int32)
        out = {pol: e.step(a) for pol, e in envs.items()}
        for pol in ("cached", "stream"):
            assert np.array_equal(out[None][0], out[pol][0]) and np.array_equal(out[None][1], out[pol][1]), (pol, t)
        if orc is not None:
            o2, r2, d2 = orc.step_autoreset(a, "next_step")
            assert np.array_equal(out[None][0], o2) and np.array_equal(out[None][1], r2), t
    for e in envs.values():
        e.close()


@pytest.mark.parametrize("case", range(24))
def test_generic_kernel_matches_oracle_on_random_shapes(case):
    """The generic (every shape at run time) kernel on 24 random warehouses — rows / columns / column height, 1..12 agents,
    queue length, sensor range 1..5, reward type, inactivity limit, normalised coordinates, ragged batches and odd launch
    geometries — against the oracle on every env, per-step launches across autoresets and a fused rollout."""
    g = np.random.default_rng(9000 + case)
    rows, cols, height = int(g.integers(1, 5)), int(g.choice([3, 5, 7])), int(g.integers(1, 10))
    n_agents = int(g.integers(1, 13))
    kw = dict(shelf_columns=cols, column_height=height, shelf_rows=rows, n_agents=n_agents, msg_bits=0,
              sensor_range=int(g.integers(1, 6)), request_queue_size=int(g.integers(0, min(2 * n_agents, rows * cols * height // 2) + 1)),
              max_inactivity_steps=(None if g.random() < 0.6 else int(g.integers(8, 25))), max_steps=int(g.integers(20, 60)),
              reward_type=int(g.integers(0, 3)), normalised_coordinates=bool(g.random() < 0.25))
    B = int(g.integers(40, 400))
    mode = ["next_step", "same_step"][case % 2]
    E, T = int(g.choice([4, 8, 12])), int(g.choice([64, 128, 256]))
    env = rware_amd.WarehouseVecEnv(B, envs_per_workgroup=E, threads_per_workgroup=T, autoreset_mode=mode, **kw)
    orc = OracleVecEnv(B, **kw)
    assert np.array_equal(env.reset(seed=90 + case)[0], orc.reset(seed=90 + case))
    rng = np.random.default_rng(case)
    for t in range(90):
        a = rng.choice(5, size=(B, n_agents), p=[.1, .55, .1, .1, .15]).astype(np.int32)
        obs, rew, term, _, _ = env.step(a)
        o2, r2, d2 = orc.step_autoreset(a, mode)
        assert np.array_equal(obs, o2) and np.array_equal(rew, r2) and np.array_equal(term, d2.astype(bool)), (t, kw, B, E, T)
    acts = rng.choice(5, size=(30, B, n_agents), p=[.1, .55, .1, .1, .15]).astype(np.int32)
    obs, rew, term = env.rollout(acts)
    for k in range(30):
        o2, r2, d2 = orc.step_autoreset(acts[k], mode)
        assert np.array_equal(obs[k], o2) and np.array_equal(rew[k], r2) and np.array_equal(term[k], d2.astype(bool)), (k, kw)
    st, so = env.get_state(), orc.get_state()
    for k in so:
        assert np.array_equal(st[k], so[k]), (k, kw)
    env.close()


def test_capture_loop_replays_policy_and_step_from_one_graph():
    """capture_loop: (policy, step) rounds in ONE HIP graph — replayed, they produce what the same rounds produce eagerly."""
    import torch
    B, N, K, R = 2048, 4, 5, 12
    kw = rware_amd.env_kwargs("rware-small-4ag-v1")
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        genv = rware_amd.WarehouseVecEnv(B, output="torch", **kw)
        eenv = rware_amd.WarehouseVecEnv(B, output="torch", **kw)
        og, _ = genv.reset(seed=3)
        oe, _ = eenv.reset(seed=3)
        W = torch.randn(og.shape[-1], 5, device="cuda")

        def policy(obs, rew, term):
            return (obs.view(-1, obs.shape[-1]) @ W).argmax(-1).view(B, N)      # int64, as a real policy hands it over

        loop = genv.capture_loop(policy, steps=K)
        assert loop.steps == K
        for r in range(R):
            loop.replay()
            for k in range(K):
                oe, re_, te, _, _ = eenv.step(policy(oe, None, None))
            if r in (0, 3):
                # a replay runs without host code: the derived views (grid, agent_* arrays) must still come back current,
                # and a host write of ONE agent view behind a replay must re-pack the live records, not stale ones
                a, b = genv.get_state(), eenv.get_state()
                for k in a:
                    assert np.array_equal(a[k], b[k]), (k, r)
                if r == 3:
                    for e_ in (genv, eenv):
                        e_.set_state(agent_dir=(a["agent_dir"] + 1) % 4)
                    oe = eenv.observations()
                    a, b = genv.get_state(), eenv.get_state()
                    for k in a:
                        assert np.array_equal(a[k], b[k]), (k, "after set_state")
        torch.cuda.synchronize()
        genv.sync(); eenv.sync()
        for name in ("obs", "rewards", "terminated"):
            assert torch.equal(genv.device_tensor(name), eenv.device_tensor(name)), name
    a, b = genv.get_state(), eenv.get_state()
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    assert a["steps"].max() == K * R
    genv.close(); eenv.close()
    # an env built on the DEFAULT stream (the usual case): capture_loop moves it to a stream of its own and replay() bridges
    # the two streams with event waits — the caller's default-stream ops before and after stay ordered, no host sync
    denv = rware_amd.WarehouseVecEnv(B, output="torch", **kw)
    eenv = rware_amd.WarehouseVecEnv(B, output="torch", **kw)
    od, _ = denv.reset(seed=4)
    oe, _ = eenv.reset(seed=4)
    loop = denv.capture_loop(policy, steps=K)
    seen = []
    for r in range(R):
        loop.replay()
        seen.append(od.sum())                                   # a default-stream consumer right behind the replay
        for k in range(K):
            oe, _, _, _, _ = eenv.step(policy(oe, None, None))
        assert torch.equal(seen[-1], oe.sum()), r
    torch.cuda.synchronize()
    a, b = denv.get_state(), eenv.get_state()
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    denv.close(); eenv.close()


@pytest.mark.parametrize("name,tile", [("layoutstr-3ag", 16), ("sr5-12ag-colheight5-twostage", 4), ("small-3ag-normcoord-sr3", 8),
                                        ("img-tiny-3ag-northup-sr2", 8), ("msg3-tiny-3ag-sr2", 8), ("small-19ag", 4)])
def test_runtime_specialised_builds_replay_reference_golden(name, tile, tmp_path, monkeypatch):
    """Run-time specialisation (hipRTC, rware_jit.cpp): shapes without an ahead-of-time exact-shape kernel — a `layout=` string,
    column_height 5 with sensor_range 5, normalised coordinates with sensor_range 3, an IMAGE / a message variant with
    sensor_range 2 — are compiled by rw_create as rw::StaticCfg builds (`build_kind == 1`) and replay the unmodified reference's
    golden traces in full; a second engine of the same shape takes the code object from the disk cache."""
    monkeypatch.setenv("RWARE_JIT_CACHE", str(tmp_path))
    meta, z = gu.load_fixture(name)
    be = EngineBackend(meta["E"], tile=tile, jit="force", **gu.ctor_kwargs(meta))
    info = be.env.engines[0].info
    assert info.jit == 1 and info.build_kind == 1 and info.specialised == 1, be.env.engines[0].jit_log()
    assert gu.replay(be, meta, z) == meta["T"]
    be.env.close()
    again = EngineBackend(meta["E"], tile=tile, jit="force", **gu.ctor_kwargs(meta))
    assert again.env.engines[0].info.jit == 2, again.env.engines[0].jit_log()     # from the cache
    again.env.close()


def test_corrupt_cached_code_object_is_dropped_and_recompiled(tmp_path, monkeypatch):
    """A cached code object the runtime refuses to load (junk behind a well-formed cache header — a truncated copy, a disk error):
    rw_create drops the file, compiles the shape afresh, runs the specialised build and says what happened; the NEXT construction finds
    a good cache again.  (rware_jit.cpp's own handling of malformed cache files runs under ASAN in oracle/sanitize.sh.)"""
    monkeypatch.setenv("RWARE_JIT_CACHE", str(tmp_path))
    kw = dict(rware_amd.env_kwargs("rware-tiny-2ag-v1"), n_agents=3, request_queue_size=3, column_height=5, max_steps=30)
    okw = dict(kw, reward_type=rware_amd.enums.enum_value(kw["reward_type"]))
    B = 512
    env = rware_amd.WarehouseVecEnv(B, jit="force", **kw)
    assert env.engines[0].info.jit == 1, env.engines[0].jit_log()
    env.close()
    files = [f for f in os.listdir(tmp_path) if f.endswith(".hsaco")]
    assert len(files) == 1
    path = os.path.join(tmp_path, files[0])
    blob = open(path, "rb").read()
    head = blob.split(b"\n", 3)
    assert head[0] == b"RWJIT1" and len(head) == 4
    with open(path, "wb") as f:   # the header stays, the code object becomes noise of the same length
        f.write(b"\n".join(head[:3]) + b"\n" + np.random.default_rng(1).integers(0, 256, size=len(head[3]), dtype=np.uint8).tobytes())
    env = rware_amd.WarehouseVecEnv(B, jit="force", **kw)
    eng = env.engines[0]
    log = eng.jit_log()
    assert eng.info.jit == 1 and eng.info.build_kind == 1, log                      # compiled afresh, not the generic kernel
    assert "loading the code object failed" in log and "dropped the cached file" in log and "retry" in log, log
    orc = OracleVecEnv(B, **okw)
    assert np.array_equal(env.reset(seed=5)[0], orc.reset(seed=5))
    rng = np.random.default_rng(6)
    for t in range(40):
        a = rng.integers(0, 5, size=(B, 3)).astype(np.int32)
        obs, rew, term, _, _ = env.step(a)
        o2, r2, d2 = orc.step_autoreset(a, "next_step")
        assert np.array_equal(obs, o2) and np.array_equal(rew, r2) and np.array_equal(term, d2.astype(bool)), t
    env.close()
    env = rware_amd.WarehouseVecEnv(B, jit="force", **kw)
    assert env.engines[0].info.jit == 2, env.engines[0].jit_log()                    # the cache is good again
    env.close()


def test_runtime_specialisation_policy(tmp_path, monkeypatch):
    """Default policy: a shape without an exact / agent-count-static build is specialised at construction when the batch has at
    least 4096 envs; small batches, registered shapes and jit=False keep the ahead-of-time kernels.  The specialised build
    gives the generic kernel's results (both against the oracle)."""
    monkeypatch.setenv("RWARE_JIT_CACHE", str(tmp_path))
    kw = dict(rware_amd.env_kwargs("rware-small-4ag-v1"), column_height=5, sensor_range=2, max_steps=40)
    small = rware_amd.WarehouseVecEnv(256, **kw)
    assert small.engines[0].info.jit == 0 and small.engines[0].info.build_kind == 0        # generic: the batch is small
    small.close()
    reg = rware_amd.WarehouseVecEnv(4096, **rware_amd.env_kwargs("rware-small-4ag-v1"))
    assert reg.engines[0].info.jit == 0 and reg.engines[0].info.build_kind == 1            # ahead-of-time exact build
    reg.close()
    # (16384 envs: four workgroups per CU — a run-time compiled build once left its stage-in DMA in flight across the barrier,
    #  hipRTC's __syncthreads() not implying the vmcnt wait hipcc's does; one workgroup per CU hid it, a full chip did not)
    B, N = 16384, kw["n_agents"]
    off = rware_amd.WarehouseVecEnv(B, jit=False, **kw)
    assert off.engines[0].info.jit == 0 and off.engines[0].info.build_kind == 0
    env = rware_amd.WarehouseVecEnv(B, **kw)
    assert env.engines[0].info.jit == 1 and env.engines[0].info.build_kind == 1, env.engines[0].jit_log()
    okw = dict(kw, reward_type=rware_amd.enums.enum_value(kw["reward_type"]))
    orc = OracleVecEnv(B, **okw)
    assert np.array_equal(env.reset(seed=12)[0], orc.reset(seed=12))
    off.reset(seed=12)
    rng = np.random.default_rng(13)
    for t in range(90):
        a = rng.choice(5, size=(B, N), p=[.1, .5, .1, .1, .2]).astype(np.int32)
        obs, rew, term, _, _ = env.step(a)
        ob2, rw2, tm2, _, _ = off.step(a)
        o2, r2, d2 = orc.step_autoreset(a, "next_step")
        assert np.array_equal(obs, o2) and np.array_equal(rew, r2) and np.array_equal(term, d2.astype(bool)), t
        assert np.array_equal(ob2, o2) and np.array_equal(rw2, r2), t
    acts = rng.choice(5, size=(20, B, N), p=[.1, .5, .1, .1, .2]).astype(np.int32)
    obs, rew, term = env.rollout(acts)                       # the specialised fused-rollout kernel
    for k in range(20):
        o2, r2, d2 = orc.step_autoreset(acts[k], "next_step")
        assert np.array_equal(obs[k], o2) and np.array_equal(rew[k], r2) and np.array_equal(term[k], d2.astype(bool)), k
    st, so = env.get_state(), orc.get_state()
    for k in so:
        assert np.array_equal(st[k], so[k]), k
    env.close(); off.close()
    # the run-time compiled build of a REGISTERED shape at 16 workgroups per CU: the same kernel source as the ahead-of-time
    # build next to it, every observation of every env
    kw4 = rware_amd.env_kwargs("rware-small-4ag-v1")
    B = 65536
    jenv, aenv = rware_amd.WarehouseVecEnv(B, jit=True, **kw4), rware_amd.WarehouseVecEnv(B, **kw4)
    assert jenv.engines[0].info.jit in (1, 2) and aenv.engines[0].info.jit == 0
    # 4096 workgroups = two rounds of 8 per CU: both launches run at raised wavefront priority instead of a start stagger (round 6)
    assert (aenv.engines[0].info.stagger_ticks, jenv.engines[0].info.stagger_ticks) == (0, 0)
    assert (aenv.engines[0].info.wave_priority, jenv.engines[0].info.wave_priority) == (3, 3)
    orc = OracleVecEnv(B, **dict(kw4, reward_type=rware_amd.enums.enum_value(kw4["reward_type"])))
    o0 = orc.reset(seed=3)
    assert np.array_equal(jenv.reset(seed=3)[0], o0) and np.array_equal(aenv.reset(seed=3)[0], o0)
    for t in range(12):
        a = rng.integers(0, 5, size=(B, 4), dtype=np.int32)
        o2 = orc.step_autoreset(a, "next_step")[0]
        assert np.array_equal(jenv.step(a)[0], o2) and np.array_equal(aenv.step(a)[0], o2), t
    jenv.close(); aenv.close()


@pytest.mark.parametrize("threads", ["0", "1"])
def test_rw_multi_one_call_steps_eight_engines(threads, monkeypatch):
    """rw_multi ("a single C call that fans out"): 8 engines — here all on this box's one GPU, one per device
    on a node — stepped by ONE C call per round, launcher thread per engine.  Same results as stepping them one call each; and
    one C call per round costs the host less than eight Python -> ctypes calls."""
    import time
    import torch
    monkeypatch.setenv("RWARE_MULTI_THREADS", threads)   # "1": the launcher-thread mode of a real multi-GPU node, forced onto one device
    kw = rware_amd.env_kwargs("rware-small-4ag-v1")
    n, Bs = 8, 1024
    mk = lambda: rware_amd.WarehouseVecEnv(n * Bs, devices=[0] * n, output="torch", **kw)
    a, b = mk(), mk()
    a.reset(seed=9); b.reset(seed=9)
    acts = torch.randint(0, 5, (64, n, Bs, 4), dtype=torch.int32, device="cuda")
    rounds = 300
    for t in range(rounds):
        a.step([acts[t % 64, k] for k in range(n)])                # -> rw_multi_step_device
        for k, eng in enumerate(b.engines):
            eng.step_device(acts[t % 64, k].data_ptr())            # one C call per engine
    torch.cuda.synchronize()
    a.sync(); b.sync()
    sa, sb = a.get_state(), b.get_state()
    for k in sa:
        assert np.array_equal(sa[k], sb[k]), k
    # host time per round: the multi call vs ONE single-engine call (pre-built pointer lists: the C calls themselves)
    ptrs = [[acts[t, k].data_ptr() for k in range(n)] for t in range(64)]
    multi, e0 = a._multi, b.engines[0]
    def per_round(fn, reps=2000):
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for t in range(reps):
                fn(t % 64)
            best = min(best, (time.perf_counter() - t0) / reps * 1e6)
            torch.cuda.synchronize()
        return best
    us_multi = per_round(lambda t: multi.step_device(ptrs[t]))
    us_single = per_round(lambda t: e0.step_device(ptrs[t][0]))
    us_loop = per_round(lambda t: [eng.step_device(p) for eng, p in zip(b.engines, ptrs[t])], reps=500)
    print(f"host us per round: rw_multi x8 {us_multi:.2f}, one rw_step_device {us_single:.2f}, eight calls in a loop {us_loop:.2f}")
    # (all eight engines share this box's one device, so the call loops over them itself — launcher threads are for engines on
    #  devices of their own: what it saves here is seven Python -> ctypes round trips)
    if threads == "0":   # (one device is one submission queue: the margin over eight Python calls is 1 .. 3 us of ~30 and not stable from
        assert us_multi < 1.2 * us_loop, (us_multi, us_single, us_loop)   # box to box — 31.8 vs 28.7 once: the claim checked is "not worse")
    a.close(); b.close()


def test_start_stagger_only_for_launches_of_two_or_more_rounds(monkeypatch):
    """rw_info.stagger_ticks: 0 at the headline batch (1024 workgroups, 4 per CU) and at a full single round (2048); from two rounds on
    25 where the launch does not run at raised wavefront priority (round 6: with the priority — every step under 200 MB of observations —
    the delay is only a delay), 0 where it does; RWARE_STAGGER_TICKS moves it.  With and without the stagger the same observations,
    rewards and state."""
    kw = rware_amd.env_kwargs("rware-small-4ag-v1")
    for B, want in ((16384, 0), (32768, 0), (65536, 0), (262144, 25)):
        env = rware_amd.WarehouseVecEnv(B, **kw)
        assert env.engines[0].info.stagger_ticks == want, B
        assert env.engines[0].info.wave_priority & 1 == (0 if want else 1), B
        env.close()
    monkeypatch.setenv("RWARE_PRIO", "0")     # (the rule without the priority: round 4's)
    a = rware_amd.WarehouseVecEnv(65536, **kw)
    monkeypatch.setenv("RWARE_STAGGER_TICKS", "0")
    b = rware_amd.WarehouseVecEnv(65536, **kw)
    assert a.engines[0].info.stagger_ticks == 25 and b.engines[0].info.stagger_ticks == 0
    a.reset(seed=4); b.reset(seed=4)
    rng = np.random.default_rng(5)
    for t in range(40):
        act = rng.integers(0, 5, size=(65536, 4), dtype=np.int32)
        oa, ra, ta, _, _ = a.step(act)
        ob, rb, tb, _, _ = b.step(act)
        assert np.array_equal(oa, ob) and np.array_equal(ra, rb) and np.array_equal(ta, tb), t
    sa, sb = a.get_state(), b.get_state()
    for k in sa:
        assert np.array_equal(sa[k], sb[k]), k
    a.close(); b.close()


def test_start_stagger_rule_for_13_to_16_agents_matches_oracle():
    """Round 6: launches of 13 .. 16 agents stagger their workgroups' starts by wider slots, already when the launch is resident at once
    (one round) and from four rounds on, not at exactly two; sensor_range 2 (BASELINE config 5) runs at raised wavefront priority
    instead (second session: 33.9 against 34.3 us with the 40 ticks it had without the priority).  The rule as rw_info shows it, and —
    a delay, never a different result — the staggered launches against the oracle."""
    for env_id, extra, B, want in (("rware-large-16ag-v1", {}, 16384, 55), ("rware-large-16ag-v1", {}, 32768, 0), ("rware-small-14ag-v1", {}, 65536, 55),
                                   ("rware-large-16ag-v1", {"sensor_range": 2}, 16384, 0), ("rware-large-16ag-v1", {"sensor_range": 2}, 32768, 0),
                                   ("rware-small-12ag-v1", {}, 16384, 0), ("rware-small-17ag-v1", {}, 16384, 0)):
        env = rware_amd.WarehouseVecEnv(B, **dict(rware_amd.env_kwargs(env_id), **extra))
        assert env.engines[0].info.stagger_ticks == want, (env_id, extra, B)
        env.close()
    for env_id, B in (("rware-medium-13ag-v1", 16384), ("rware-large-16ag-v1", 16384)):
        kw = rware_amd.env_kwargs(env_id)
        kw["reward_type"] = rware_amd.enums.enum_value(kw["reward_type"])
        kw["max_steps"] = 20
        env = rware_amd.WarehouseVecEnv(B, **kw)
        assert env.engines[0].info.stagger_ticks == 55
        orc = OracleVecEnv(B, **kw)
        assert np.array_equal(env.reset(seed=9)[0], orc.reset(seed=9))
        rng = np.random.default_rng(2)
        for t in range(45):
            a = rng.choice(5, size=(B, kw["n_agents"]), p=[0.1, 0.55, 0.1, 0.1, 0.15]).astype(np.int32)
            obs, rew, term, _, _ = env.step(a)
            o2, r2, d2 = orc.step_autoreset(a, "next_step")
            assert np.array_equal(obs, o2) and np.array_equal(rew, r2) and np.array_equal(term, d2.astype(bool)), (env_id, t)
        st, so = env.get_state(), orc.get_state()
        assert all(np.array_equal(st[k], so[k]) for k in so)
        env.close()


def test_wave_priority_rule_and_results(monkeypatch):
    """Round 6: rw_info.wave_priority — the per-step launches run their dependent chain (stage-in, agent phases) at wavefront priority 3
    and drop to 0 behind the agent-phase barrier.  The rule as rw_info shows it; with the priority forced on and off the same
    observations, rewards, flags and state on the shapes of both families (a scheduling hint, never a different result)."""
    for env_id, extra, B, want in (("rware-small-4ag-v1", {}, 16384, 1), ("rware-tiny-2ag-v1", {}, 4096, 1), ("rware-medium-6ag-hard-v1", {}, 8192, 1),
                                   ("rware-large-16ag-v1", {"sensor_range": 2}, 16384, 1), ("rware-small-12ag-v1", {}, 16384, 1),
                                   ("rware-medium-13ag-v1", {}, 16384, 0), ("rware-large-16ag-v1", {}, 16384, 0), ("rware-small-17ag-v1", {}, 16384, 1),
                                   ("rware-small-4ag-v1", {}, 131072, 1), ("rware-small-4ag-v1", {}, 262144, 0), ("rware-small-12ag-v1", {}, 65536, 0)):
        env = rware_amd.WarehouseVecEnv(B, **dict(rware_amd.env_kwargs(env_id), **extra))
        assert env.engines[0].info.wave_priority & 1 == want, (env_id, extra, B)          # bit 0: the per-step launches
        assert (env.engines[0].info.wave_priority >> 1) == (0 if B * rware_amd.env_kwargs(env_id)["n_agents"] * env.obs_length * 4 > 200e6 else 1)   # bit 1: the fused rollouts
        env.close()
    for env_id, extra, B in (("rware-small-4ag-v1", {}, 32768), ("rware-medium-13ag-v1", {}, 16384), ("rware-large-16ag-v1", {"sensor_range": 2}, 8192)):
        kw = dict(rware_amd.env_kwargs(env_id), max_steps=17, **extra)
        monkeypatch.setenv("RWARE_HOOKS", "1")
        monkeypatch.setenv("RWARE_PRIO", "1"); monkeypatch.setenv("RWARE_PRIO_ROLLOUT", "1")
        a = rware_amd.WarehouseVecEnv(B, **kw)
        monkeypatch.setenv("RWARE_PRIO", "0"); monkeypatch.setenv("RWARE_PRIO_ROLLOUT", "0")
        b = rware_amd.WarehouseVecEnv(B, **kw)
        monkeypatch.delenv("RWARE_PRIO"); monkeypatch.delenv("RWARE_PRIO_ROLLOUT")
        assert (a.engines[0].info.wave_priority, b.engines[0].info.wave_priority) == (3, 0)
        a.reset(seed=4); b.reset(seed=4)
        rng = np.random.default_rng(5)
        for t in range(40):
            act = rng.choice(5, size=(B, kw["n_agents"]), p=[0.1, 0.55, 0.1, 0.1, 0.15]).astype(np.int32)
            ra, rb = a.step(act), b.step(act)
            assert all(np.array_equal(x, y) for x, y in zip(ra[:4], rb[:4])), (env_id, t)
        tape = rng.choice(5, size=(12, B, kw["n_agents"]), p=[0.1, 0.55, 0.1, 0.1, 0.15]).astype(np.int32)
        fa, fb = a.rollout(tape, want_obs=False), b.rollout(tape, want_obs=False)        # the fused rollouts, with and without
        assert all(np.array_equal(x, y) for x, y in zip(fa[1:], fb[1:])), env_id
        sa, sb = a.get_state(), b.get_state()
        assert all(np.array_equal(sa[k], sb[k]) for k in sa), env_id
        a.close(); b.close()


def test_pipelines_of_small_sub_batches_run_without_the_wavefront_priority():
    """make_pipelines: with two launches in flight the raised chain of one takes issue slots from the other's store phase, so sub-batches of
    8192 envs or fewer are built with RW_PRIO_OFF (profiles/r06_pipelines_prio.txt); bigger ones keep the engine's rule; the caller's
    wave_priority= wins."""
    def priorities(B, **kw):
        pipes = rware_amd.make_pipelines(B, 2, env_id="rware-small-10ag-v1", **kw)
        out = [p.env.engines[0].info.wave_priority for p in pipes]
        for p in pipes:
            p.env.close()
        return out
    assert priorities(16384) == [0, 0]
    assert priorities(32768) == [3, 3]
    assert priorities(16384, wave_priority=True) == [3, 3]
    one = rware_amd.WarehouseVecEnv(8192, **rware_amd.env_kwargs("rware-small-10ag-v1"))
    assert one.engines[0].info.wave_priority == 3
    one.close()


def test_13_to_16_agents_step_on_4_env_workgroups_and_roll_out_on_8(monkeypatch):
    """Round 6: the per-step launches of 13 .. 16 agents (sensor_range 1) run on 4-env workgroups at raised wavefront priority below one
    full round of 8-env workgroups (16384 envs) and between one and four rounds, the fused rollouts on the 8-env build at every batch
    (profiles/r06_1316_matrix.txt, r06_1316_rollout_geom.txt).  The rule as rw_info shows it, and the two kernels of one engine — different
    launch geometries — against the oracle, per-step launches and fused rollouts interleaved."""
    for env_id, B, want in (("rware-large-16ag-v1", 4096, (4, 3, 0)), ("rware-large-16ag-v1", 8192, (4, 3, 0)), ("rware-large-16ag-v1", 16384, (8, 2, 55)),
                            ("rware-large-16ag-v1", 32768, (4, 3, 0)), ("rware-large-16ag-v1", 65536, (8, 0, 55)), ("rware-medium-13ag-v1", 49152, (4, 3, 0)),   # (65536 x 16 agents: 297 MB of observations per step, past the priority's size limit)
                            ("rware-small-14ag-v1", 24576, (4, 3, 0)), ("rware-tiny-14ag-v1", 4096, (8, 3, 0)), ("rware-tiny-14ag-v1", 16384, (8, 2, 55)),
                            # 9 .. 12 and 17 .. 19 agents: the 4-env build below 8192 envs (its workgroups stay under one round), the 8-env one from there on
                            ("rware-small-12ag-v1", 4096, (4, 3, 0)), ("rware-small-10ag-v1", 6144, (4, 3, 0)), ("rware-small-10ag-v1", 8192, (8, 3, 0)),
                            ("rware-small-19ag-v1", 4096, (4, 3, 0)), ("rware-small-17ag-v1", 32768, (8, 3, 0)), ("rware-tiny-10ag-v1", 4096, (8, 3, 0)),
                            ("rware-small-8ag-v1", 4096, (8, 3, 0))):
        env = rware_amd.WarehouseVecEnv(B, **rware_amd.env_kwargs(env_id))
        i = env.engines[0].info
        assert (i.envs_per_workgroup, i.wave_priority, i.stagger_ticks) == want, (env_id, B)
        env.close()
    monkeypatch.setenv("RWARE_WIDE_E4", "0")
    env = rware_amd.WarehouseVecEnv(4096, **rware_amd.env_kwargs("rware-large-16ag-v1"))
    assert env.engines[0].info.envs_per_workgroup == 8
    env.close()
    monkeypatch.delenv("RWARE_WIDE_E4")
    # (9 .. 12 agents: the fused rollout follows onto the 4-env build; 13 .. 19: it stays on the 8-env one)
    for env_id, B in (("rware-large-16ag-v1", 4096), ("rware-medium-13ag-v1", 32768), ("rware-small-15ag-v1", 8192), ("rware-small-10ag-v1", 4096),
                      ("rware-small-19ag-v1", 6144)):
        kw = rware_amd.env_kwargs(env_id)
        kw["reward_type"] = rware_amd.enums.enum_value(kw["reward_type"])
        kw["max_steps"] = 17
        N = kw["n_agents"]
        env = rware_amd.WarehouseVecEnv(B, **kw)
        assert env.engines[0].info.envs_per_workgroup == 4
        orc = OracleVecEnv(B, **kw)
        assert np.array_equal(env.reset(seed=9)[0], orc.reset(seed=9))
        rng = np.random.default_rng(2)
        for rnd in range(2):
            for t in range(12):
                a = rng.choice(5, size=(B, N), p=[0.1, 0.55, 0.1, 0.1, 0.15]).astype(np.int32)
                obs, rew, term, _, _ = env.step(a)
                o2, r2, d2 = orc.step_autoreset(a, "next_step")
                assert np.array_equal(obs, o2) and np.array_equal(rew, r2) and np.array_equal(term, d2.astype(bool)), (env_id, rnd, t)
            tape = rng.choice(5, size=(8, B, N), p=[0.1, 0.55, 0.1, 0.1, 0.15]).astype(np.int32)
            _, rew, term = env.rollout(tape, want_obs=False)
            for t in range(8):
                _, r2, d2 = orc.step_autoreset(tape[t], "next_step")
                assert np.array_equal(rew[t], r2) and np.array_equal(term[t], d2.astype(bool)), (env_id, rnd, t)
        st, so = env.get_state(), orc.get_state()
        assert all(np.array_equal(st[k], so[k]) for k in so), env_id
        env.close()


def test_two_pipelines_on_one_device_match_the_single_engine():
    """Double-buffered sampling (bench.py's `two_pipelines`): the batch as two engines on one device, own streams, stepped
    CONCURRENTLY by one launcher thread each, against one engine over the whole batch — same observations and state."""
    import threading
    import torch
    kw = rware_amd.env_kwargs("rware-small-10ag-v1")
    B, N, T = 8192, 10, 48
    two, one = rware_amd.WarehouseVecEnv(B, devices=[0, 0], **kw), rware_amd.WarehouseVecEnv(B, **kw)
    assert len(two.engines) == 2
    two.reset(seed=21); one.reset(seed=21)
    acts = np.random.default_rng(22).choice(5, size=(T, B, N), p=[.1, .55, .1, .1, .15]).astype(np.int32)
    full = torch.from_numpy(acts).cuda()
    halves = [torch.from_numpy(np.ascontiguousarray(acts[:, k * (B // 2):(k + 1) * (B // 2)])).cuda() for k in range(2)]

    def run(k):
        two.engines[k].step_tape_device(halves[k].data_ptr(), T, 0, T)
        two.engines[k].sync()
    th = [threading.Thread(target=run, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    one.engines[0].step_tape_device(full.data_ptr(), T, 0, T)
    one.engines[0].sync()
    for t in th:
        t.join()
    for e in two.engines + one.engines:
        e.mark_views_stale()
    sa, sb = two.get_state(), one.get_state()
    for k in sb:
        assert np.array_equal(sa[k], sb[k]), k
    assert np.array_equal(two.observations(), one.observations())
    two.close(); one.close()


def test_make_pipelines_closed_loop_equals_the_whole_batch():
    """rware_amd.make_pipelines: the batch as two sub-batches on torch streams of their own, stepped alternately from one host
    thread with CUDA action tensors (nothing orders one sub-batch behind the other) — every observation, reward and terminated
    flag equal to ONE env over the whole batch, step by step."""
    import torch
    kw = rware_amd.env_kwargs("rware-small-10ag-v1")
    kw["max_steps"] = 30
    B, N, T = 4096, 10, 70
    pipes = rware_amd.make_pipelines(B, 2, **kw)
    ref = rware_amd.WarehouseVecEnv(B, **kw)
    assert [(p.lo, p.hi) for p in pipes] == [(0, 2048), (2048, 4096)] and pipes[0].stream != pipes[1].stream
    o_ref = ref.reset(seed=50)[0]
    o0 = [p.reset(seed=50)[0] for p in pipes]
    with pipes[0]:
        a0 = o0[0].cpu().numpy()
    with pipes[1]:
        a1 = o0[1].cpu().numpy()
    assert np.array_equal(np.concatenate([a0, a1]), o_ref)
    acts = np.random.default_rng(51).choice(5, size=(T, B, N), p=[.1, .55, .1, .1, .15]).astype(np.int32)
    dev = [[torch.from_numpy(np.ascontiguousarray(acts[t, p.lo:p.hi])).cuda() for t in range(T)] for p in pipes]
    torch.cuda.synchronize()
    for t in range(T):
        got = []
        for k, p in enumerate(pipes):
            with p as env:
                obs, rew, term, _, _ = env.step(dev[k][t])
                got.append((obs.cpu().numpy(), rew.cpu().numpy(), term.cpu().numpy()))   # (stream-ordered copies on the pipeline's stream)
        o2, r2, d2, _, _ = ref.step(acts[t])
        assert np.array_equal(np.concatenate([g[0] for g in got]), o2), t
        assert np.array_equal(np.concatenate([g[1] for g in got]), r2) and np.array_equal(np.concatenate([g[2] for g in got]), d2), t
    for p in pipes:
        p.env.close()
    ref.close()


def test_capture_pipelines_replays_what_the_eager_pipelines_do():
    """rware_amd.capture_pipelines: (policy, step) rounds of both pipelines in ONE HIP graph, a branch per pipeline — after the
    replay every env is where the same rounds issued eagerly (and where ONE env over the whole batch) put it."""
    import torch
    kw = rware_amd.env_kwargs("rware-small-6ag-v1")
    B, R = 2048, 24
    W = (torch.arange(71 * 5, device="cuda", dtype=torch.float32).reshape(71, 5) % 7 - 3.0) * 0.25

    def policy(obs, rew, term):
        return (obs @ W).argmax(-1).to(torch.int32)

    cap, eag = rware_amd.make_pipelines(B, 2, **kw), rware_amd.make_pipelines(B, 2, **kw)
    one = rware_amd.WarehouseVecEnv(B, output="torch", **kw)
    for p in cap + eag:
        p.reset(seed=11)
    o1 = one.reset(seed=11)[0]
    graph = rware_amd.capture_pipelines(cap, policy, steps=R)
    graph.replay()
    oe = [None, None]
    for k, p in enumerate(eag):
        with p as env:
            v = env._torch_views()
            obs, rew, term = env._obs_of(v), v["rewards"], v["terminated_bool"]
            for _ in range(R):
                obs, rew, term, _, _ = env.step(policy(obs, rew, term))
            oe[k] = obs
    rew1 = torch.zeros((B, 6), device="cuda"); term1 = torch.zeros((B,), dtype=torch.bool, device="cuda")
    for _ in range(R):
        o1, rew1, term1, _, _ = one.step(policy(o1, rew1, term1))
    torch.cuda.synchronize()
    for k in range(2):
        sc, se = cap[k].env.get_state(), eag[k].env.get_state()
        for name in se:
            assert np.array_equal(sc[name], se[name]), (k, name)
        assert torch.equal(cap[k].env._obs_of(cap[k].env._torch_views()), oe[k])
    whole = torch.cat([cap[k].env._obs_of(cap[k].env._torch_views()) for k in range(2)])
    assert torch.equal(whole, o1)
    graph.replay()                                   # a second replay continues from there, and eager steps still work after it
    with cap[0] as env:
        env.step(torch.zeros((B // 2, 6), dtype=torch.int32, device="cuda"))
    torch.cuda.synchronize()
    for p in cap + eag:
        p.env.close()
    one.close()


_HAS_PIPE = None


def _library_has_pipelined_builds() -> bool:
    """Does this librware_hip.so carry the chunk-pipelined persistent kernels?  Only a `make PIPE=1` build does (round 6: they were
    measured slower than the classic launch on every configuration, so the default library leaves them out)."""
    global _HAS_PIPE
    if _HAS_PIPE is None:
        env = rware_amd.WarehouseVecEnv(64, pipe=True, **rware_amd.env_kwargs("rware-small-4ag-v1"))
        _HAS_PIPE = env.engines[0].info.pipe_workgroups > 0
        env.close()
    return _HAS_PIPE


def test_pipe_request_on_the_default_library_runs_the_classic_kernel_and_says_so():
    """RW_PIPE_ON (pipe=True) against a library without the pipelined kernels: the classic kernel runs (bit-exact), `rw_get_info()`
    shows it and `rw_jit_log()` says why — nothing fails, nothing is silently different."""
    if _library_has_pipelined_builds():
        pytest.skip("this library was built with PIPE=1")
    kw = rware_amd.env_kwargs("rware-small-4ag-v1")
    kw["reward_type"] = rware_amd.enums.enum_value(kw["reward_type"])
    B = 256
    env = rware_amd.WarehouseVecEnv(B, pipe=True, **kw)
    eng = env.engines[0]
    assert eng.info.pipe_workgroups == 0 and eng.info.specialised == 1
    assert "make PIPE=1" in eng.jit_log()
    orc = OracleVecEnv(B, **kw)
    assert np.array_equal(env.reset(seed=3)[0], orc.reset(seed=3))
    rng = np.random.default_rng(3)
    for t in range(30):
        a = rng.integers(0, 5, size=(B, 4)).astype(np.int32)
        obs, rew, term, _, _ = env.step(a)
        o2, r2, d2 = orc.step_autoreset(a, "next_step")
        assert np.array_equal(obs, o2) and np.array_equal(rew, r2) and np.array_equal(term, d2.astype(bool)), t
    env.close()


@pytest.mark.parametrize("name,tile,grid", [("small-4ag", 20, 2), ("medium-6ag-hard", 8, 2), ("large-16ag-sr2", 6, 2), ("tiny-2ag", 40, 3)])
def test_pipelined_builds_replay_reference_golden(name, tile, grid, monkeypatch):
    """The reference's golden traces on the chunk-pipelined persistent builds (opt-in, pipe=True): a handful of persistent workgroups
    (RWARE_PIPE_GRID) walk several chunks each — uneven shares included — so the two-buffer hand-over is what is being replayed.
    (Needs a `make PIPE=1` library; the host-thread emulation build always has the pipelined kernels: tests/test_engine_emulated.py.)"""
    if not _library_has_pipelined_builds():
        pytest.skip("librware_hip.so was built without the pipelined kernels (make PIPE=1): quarantined, measured slower everywhere")
    monkeypatch.setenv("RWARE_PIPE_GRID", str(grid))
    meta, z = gu.load_fixture(name)
    be = EngineBackend(meta["E"], tile=tile, pipe=True, **gu.ctor_kwargs(meta))
    info = be.env.engines[0].info
    assert info.pipe_workgroups == grid and (meta["E"] * tile) // info.pipe_envs_per_workgroup > grid
    assert gu.replay(be, meta, z) == meta["T"]
    be.env.close()


@pytest.mark.parametrize("env_id,extra,B,T,mode", [
    ("rware-small-4ag-v1", {"max_steps": 60}, 65536, 130, "next_step"),              # 4096 chunks on ~1500 persistent workgroups
    ("rware-small-4ag-v1", {"max_steps": 40}, 16384, 90, "same_step"),               # one chunk per workgroup: the pipeline never fills
    ("rware-medium-6ag-hard-v1", {"max_steps": 50, "reward_type": 0}, 32768, 110, "next_step"),
    ("rware-large-16ag-v1", {"sensor_range": 2, "max_steps": 30}, 16384, 70, "same_step"),   # BASELINE config 5's shard
    ("rware-small-10ag-v1", {"max_steps": 40}, 16384, 90, "next_step"),              # per-cell agent phases, run-time queue length
    ("rware-large-16ag-v1", {"max_steps": 30}, 16384, 70, "next_step"),
])
def test_pipelined_builds_match_oracle_full_batch(env_id, extra, B, T, mode):
    kw = rware_amd.env_kwargs(env_id)
    kw.update(extra)
    kw["reward_type"] = rware_amd.enums.enum_value(kw["reward_type"])
    if not _library_has_pipelined_builds():
        pytest.skip("librware_hip.so was built without the pipelined kernels (make PIPE=1): quarantined, measured slower everywhere")
    env = rware_amd.WarehouseVecEnv(B, autoreset_mode=mode, pipe=True, **kw)
    assert env.engines[0].info.pipe_workgroups > 0
    orc = OracleVecEnv(B, **kw)
    obs, _ = env.reset(seed=11)
    assert np.array_equal(obs, orc.reset(seed=11))
    rng = np.random.default_rng(5)
    for t in range(T):
        a = rng.choice(5, size=(B, kw["n_agents"]), p=[0.1, 0.55, 0.1, 0.1, 0.15]).astype(np.int32)
        obs, rew, term, trunc, info = env.step(a)
        o2, r2, d2 = orc.step_autoreset(a, mode)
        assert np.array_equal(rew, r2) and np.array_equal(term, d2.astype(bool)), t
        assert np.array_equal(obs, o2), t
        if mode == "same_step" and d2.any():
            assert np.array_equal(info["final_obs"][orc.final_mask], orc.final_obs[orc.final_mask]), t
    st, so = env.get_state(), orc.get_state()
    for k in so:
        assert np.array_equal(st[k], so[k]), k
    env.close()


def test_library_selftest_guards_run_on_this_gpu(monkeypatch):
    """rw_selftest: the two compiler / hardware facts the kernels are written around — gathered values compared in registers of their
    own (a DPP move folded into a subtract comes out with swapped operands on gfx950) and an LDS-DMA stage-in waited for explicitly
    in front of the barrier — checked on THIS device by code prebuilt into the library (no hipcc needed, nothing to skip).  And the
    guard can fail: with the known-bad forms switched in (RWARE_SELFTEST_BREAK=1) it must."""
    from rware_amd import _capi
    ok, msg = _capi.selftest(0)
    assert ok, msg
    print("rw_selftest:", msg)
    monkeypatch.setenv("RWARE_SELFTEST_BREAK", "1")
    ok, msg = _capi.selftest(0)
    assert not ok, f"the broken forms passed: {msg}"
    print("rw_selftest (broken forms):", msg)


@pytest.mark.parametrize("env_id,extra,B", [
    ("rware-small-4ag-v1", dict(observation_type=2, max_steps=40), 4096),                                   # exact IMAGE build
    ("rware-medium-6ag-hard-v1", dict(observation_type=3, max_steps=35, max_inactivity_steps=20), 2048),     # IMAGE_DICT, generic kernel
    ("rware-tiny-2ag-v1", dict(observation_type=3, image_observation_directional=False, sensor_range=2, max_steps=30), 1000),
])
def test_image_terminal_observations_same_step(env_id, extra, B):
    """SAME_STEP autoreset keeps the terminating step's own IMAGE / IMAGE_DICT observation (rware/warehouse.py:
    527-596, 929-946) as info["final_obs"]; compared with the oracle's pre-reset image on every env that ended an episode."""
    kw = rware_amd.env_kwargs(env_id)
    kw.update(extra)
    kw["reward_type"] = rware_amd.enums.enum_value(kw["reward_type"])
    env = rware_amd.WarehouseVecEnv(B, autoreset_mode="same_step", **kw)
    orc = OracleVecEnv(B, **kw)
    assert gu.check_same_step_image_run(env, orc, B, kw["n_agents"], steps=100, seed=4) >= 2 * B
    env.close()


def test_image_terminal_observations_same_step_transposed_layers():
    kw = dict(shelf_columns=3, column_height=3, shelf_rows=2, n_agents=5, msg_bits=0, sensor_range=2, request_queue_size=3,
              max_inactivity_steps=None, max_steps=25, reward_type=1, observation_type=2, image_observation_layers=[3, 4, 0, 1, 2, 5, 6])
    env = rware_amd.WarehouseVecEnv(2048, autoreset_mode="same_step", **kw)
    orc = OracleVecEnv(2048, **kw)
    assert gu.check_same_step_image_run(env, orc, 2048, 5, steps=60, seed=9) >= 2 * 2048
    env.close()


@pytest.mark.parametrize("env_id,extra,B,T,shards", [
    ("rware-medium-6ag-hard-v1", {}, 65536, 540, 16),                 # BASELINE config 4 at its FULL batch on ONE device; crosses the mass autoreset
    ("rware-large-16ag-v1", {"sensor_range": 2}, 131072, 100, 16),    # config 5 at its FULL batch: 1.54 GB of observations per step
])
def test_largest_single_gpu_configurations_every_step(env_id, extra, B, T, shards):
    """with no 8-GPU node, BASELINE configs 4 and 5 run at their FULL batch on one device — every env's rewards and
    flags, and EVERY step's observations through a per-env fixed-weight dot product (integer weights: exact in float64) computed on
    the GPU from the engine's observation tensor and on the host from the oracle's array (tests/oracle_shards.py: the C oracle on
    16 host threads); the complete state at the end."""
    import torch
    from oracle_shards import ShardedOracle
    kw = rware_amd.env_kwargs(env_id)
    kw.update(extra)
    N = kw["n_agents"]
    env = rware_amd.WarehouseVecEnv(B, output="torch", **kw)
    assert env.engines[0].info.specialised == 1
    orc = ShardedOracle(B, shards, **dict(kw, reward_type=kw["reward_type"].value))
    w = torch.from_numpy(orc.w).cuda()

    def cs(obs):   # (in slices: the float64 copy of 1.5 GB of observations would be 3 GB)
        flat = obs.view(B, -1)
        return torch.cat([flat[i:i + 16384].to(torch.float64) @ w for i in range(0, B, 16384)]).cpu().numpy()

    obs, _ = env.reset(seed=77)
    assert np.array_equal(cs(obs), orc.reset(77))
    rng = np.random.default_rng(41)
    for t in range(T):
        a = rng.choice(5, size=(B, N), p=[.1, .5, .15, .15, .1]).astype(np.int32)
        obs, rew, term, _, _ = env.step(torch.from_numpy(a).cuda())
        c2, r2, d2 = orc.step(a)
        assert np.array_equal(rew.cpu().numpy(), r2) and np.array_equal(term.cpu().numpy(), d2.astype(bool)), t
        bad = np.nonzero(cs(obs) != c2)[0]
        assert bad.size == 0, f"step {t}: the observations of {bad.size} envs differ (first: env {bad[0]})"
    st, so = env.get_state(), orc.get_state()
    for k in so:
        assert np.array_equal(st[k], so[k]), k
    orc.close()
    env.close()


def test_c_example_program_matches_the_oracle(tmp_path):
    """examples/rware_c_example.c — the C-ABI from plain C: layout arrays, rw_config, rw_create, rw_reset with seeds, rw_step with HOST
    actions, rw_read_outputs — compiled with gcc against librware_hip.so and run as its own process; its checksum over every step's
    observations, rewards and flags equals the oracle's on the same seeds and (LCG) actions, across autoresets."""
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "robotic-warehouse_amd", "csrc")
    exe = str(tmp_path / "rware_c_example")
    subprocess.check_call(["gcc", "-std=c11", "-O2", "-I", os.path.join(root, "include"), os.path.join(root, "examples", "rware_c_example.c"),
                           "-L", csrc, "-lrware_hip", f"-Wl,-rpath,{csrc}", "-o", exe])
    B, steps, seed = 96, 75, 11
    out = subprocess.run([exe, str(B), str(steps), str(seed)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    fields = dict(f.split("=", 1) for f in out.stdout.split() if "=" in f)
    assert int(fields["envs"]) == B and int(fields["obs_length"]) == 71 and int(fields["build_kind"]) == 1

    def fnv(h, a):
        for b in np.ascontiguousarray(a).view(np.uint8).ravel().tolist():
            h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
        return h

    kw = dict(rware_amd.env_kwargs("rware-tiny-2ag-v1"), max_steps=30)
    kw["reward_type"] = rware_amd.enums.enum_value(kw["reward_type"])
    orc = OracleVecEnv(B, **kw)
    h = fnv(14695981039346656037, orc.reset(seed=seed))
    lcg, reward_sum, ends = (seed * 2654435761 + 12345) & 0xFFFFFFFF, 0.0, 0
    for t in range(steps):
        a = np.empty(B * 2, np.int32)
        for i in range(B * 2):
            lcg = (lcg * 1664525 + 1013904223) & 0xFFFFFFFF
            r = (lcg >> 16) % 10
            a[i] = 1 if r < 5 else r - 5
        o2, r2, d2 = orc.step_autoreset(a.reshape(B, 2), "next_step")
        h = fnv(fnv(fnv(h, o2.astype(np.float32)), r2.astype(np.float32)), d2.astype(np.uint8))
        reward_sum += float(r2.sum()); ends += int(d2.sum())
    assert fields["checksum"] == f"{h:016x}", (fields, f"{h:016x}")
    assert float(fields["reward_sum"]) == reward_sum and int(fields["episode_ends"]) == ends and ends >= 2 * B
