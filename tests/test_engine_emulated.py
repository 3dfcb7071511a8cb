"""CPU-side tests of the PRODUCT sources (kernel + C-ABI host + Python host layer) compiled for
host threads (tests/emu): control flow, autoreset modes, masked reset, state injection, geometry
variants and ragged batches, all diffed against the oracle / the reference's golden vectors.
The gfx950 build of the same sources is checked by the -m gpu tests."""
import numpy as np
import pytest

import golden_util as gu
from engine_backend import EngineBackend, build_emu
from rware_oracle import OracleVecEnv

import rware_amd

LIB = build_emu()
pytestmark = pytest.mark.timeout(1500)   # (256 OS threads per workgroup on pthread barriers: a loaded host slows these 50x)


@pytest.mark.parametrize("name,geom", [
    ("tiny-2ag", (4, 64)),
    ("small-4ag", (16, 256)),
    ("tiny-4ag-easy-twostage", (4, 128)),
    ("small-8ag-global-inact", (4, 64)),
    ("layoutstr-3ag", (4, 64)),
    ("small-3ag-normcoord-sr3", (4, 64)),
    ("tiny-1ag-hard-q0", (4, 64)),
    ("small-19ag", (4, 64)),
    ("img-small-4ag-directional", (0, 0)),
    ("img-tiny-3ag-northup-sr2", (4, 64)),
    ("imgdict-medium-6ag-hard", (8, 128)),
    ("msg2-small-4ag", (0, 0)),
    ("msg3-tiny-3ag-sr2", (4, 64)),
    ("img-square-5ag-transposed-layers", (0, 0)),
    ("imgdict-square-5ag-transposed-northup", (4, 128)),
    ("img-msg2-tiny-3ag-8layers", (0, 0)),
    ("sr5-12ag-colheight5-twostage", (0, 0)),
    ("img-square-all7-msg1", (4, 64)),
])
def test_emulated_engine_matches_reference_golden(name, geom):
    meta, z = gu.load_fixture(name)
    be = EngineBackend(meta["E"], library=LIB, envs_per_workgroup=geom[0], threads_per_workgroup=geom[1],
                       **gu.ctor_kwargs(meta))
    assert gu.replay(be, meta, z, steps=120) > 0   # (the GPU suite replays every trace in full)
    be.env.close()


@pytest.mark.parametrize("mode", ["next_step", "same_step", "disabled"])
@pytest.mark.parametrize("env_id,extra,B,geom", [
    ("rware-tiny-2ag-v1", {"max_steps": 25}, 7, (4, 64)),           # ragged: 2 workgroups, the last partial
    ("rware-small-8ag-v1", {"max_steps": 30, "reward_type": 0, "max_inactivity_steps": 9}, 5, (4, 128)),
    ("rware-medium-6ag-hard-v1", {"max_steps": 20, "reward_type": 2}, 9, (8, 128)),
    ("rware-large-16ag-v1", {"sensor_range": 2, "max_steps": 15}, 3, (4, 64)),
    # exact-shape builds (default geometry): agent phases in registers — DPP pairs (N = 2), quads (N = 4), ds_bpermute (N = 6)
    ("rware-tiny-2ag-v1", {"max_steps": 25}, 32, (0, 0)),
    ("rware-small-4ag-v1", {"max_steps": 25, "reward_type": 2}, 8, (0, 0)),             # the 8-env build
    ("rware-small-4ag-v1", {"max_steps": 25, "max_inactivity_steps": 11}, 32, (16, 256)),
    ("rware-medium-6ag-hard-v1", {"max_steps": 20, "reward_type": 0}, 16, (0, 0)),      # the 8-env build
    ("rware-medium-6ag-hard-v1", {"max_steps": 20}, 16, (16, 256)),
    ("rware-tiny-4ag-hard-v1", {"max_steps": 20}, 16, (0, 0)),
    # exact-shape builds, the two-pass observation expansion (normalised coordinates are fractions, not byte-sized integers)
    ("rware-small-4ag-v1", {"max_steps": 25, "normalised_coordinates": True}, 16, (16, 256)),
    # exact-shape builds of the other observation kinds, stepwise + (below) fused
    ("rware-small-4ag-v1", {"max_steps": 25, "observation_type": 2}, 16, (0, 0)),
    ("rware-small-4ag-v1", {"max_steps": 25, "msg_bits": 2}, 16, (0, 0)),
])
def test_emulated_engine_matches_oracle(env_id, extra, B, geom, mode):
    kw = rware_amd.env_kwargs(env_id)
    kw.update(extra)
    kw["reward_type"] = rware_amd.enums.enum_value(kw["reward_type"])
    env = rware_amd.WarehouseVecEnv(B, autoreset_mode=mode, library=LIB, envs_per_workgroup=geom[0],
                                    threads_per_workgroup=geom[1], **kw)
    orc = OracleVecEnv(B, **kw)
    assert env.engines[0].info.specialised == (1 if geom in ((0, 0), (16, 256)) else 0)
    obs, _ = env.reset(seed=99)
    assert np.array_equal(obs, orc.reset(seed=99))
    rng = np.random.default_rng(3)
    M = kw.get("msg_bits", 0)
    for t in range(70):
        a = rng.choice(5, size=(B, kw["n_agents"]), p=[0.1, 0.55, 0.1, 0.1, 0.15]).astype(np.int32)
        if M:
            a = np.concatenate([a[..., None], rng.integers(0, 2, size=(B, kw["n_agents"], M), dtype=np.int32)], axis=-1)
        obs, rew, term, trunc, info = env.step(a)
        o2, r2, d2 = orc.step_autoreset(a, mode)
        assert np.array_equal(rew, r2) and np.array_equal(term, d2.astype(bool)), t
        assert np.array_equal(obs, o2), t
        if mode == "same_step":
            # the terminal observation of the step that ended (and reset) an episode: info["final_obs"], as Gymnasium >= 1.0 has it
            # (FLATTENED and, round 5, the IMAGE types)
            assert ("final_obs" in info) == bool(d2.any()), t
            if d2.any():
                assert np.array_equal(info["_final_obs"], orc.final_mask)
                assert np.array_equal(info["final_obs"][orc.final_mask], orc.final_obs[orc.final_mask]), t
        else:
            assert info == {}
        st, so = env.get_state(), orc.get_state()
        for k in so:
            assert np.array_equal(st[k], so[k]), (k, t)
    env.close()


@pytest.mark.parametrize("name,geom,tile", [
    ("small-4ag", (16, 256), 4),       # quad exchange (N = 4), 16-env workgroups
    ("small-4ag", (0, 0), 2),          # ... and the 8-env build picked for small batches
    ("tiny-2ag", (0, 0), 4),           # pair exchange (N = 2)
    ("medium-6ag-hard", (0, 0), 8),    # ds_bpermute exchange (N = 6), 8-env build
    ("large-16ag-sr2", (0, 0), 4),     # exact-shape build, N = 16 in registers (128-bit chain links; round 4)
    ("img-small-4ag-directional", (0, 0), 16),   # exact-shape IMAGE build
    ("msg2-small-4ag", (0, 0), 16),              # exact-shape build with 2 communication bits
    ("small-8ag-global-inact", (0, 0), 16),      # N = 8 in registers: 64-bit chain links (round 3)
    ("tiny-4ag-easy-twostage", (0, 0), 16),      # Q > N: two queue slots per agent lane
    ("small-7ag-hard", (0, 0), 4),               # agent-count-static build, N = 7 (ds_bpermute, 64-bit links), 8-env workgroups
    ("large-4ag", (0, 0), 8),                    # agent-count-static build on the large warehouse
    ("medium-2ag-easy", (32, 256), 8),           # 2 agents, the 32-env build (pair exchange), Q = 2 N
    ("small-19ag", (0, 0), 4),                   # agent-count-static build, N = 19: 128-bit chain links, three agent wavefronts
    ("small-19ag", (4, 256), 2),                 # ... and its 4-env geometry
])
def test_emulated_exact_shape_builds_match_reference_golden(name, geom, tile):
    """The golden traces of the unmodified reference, replayed on the EXACT-SHAPE kernel builds (the ones the BASELINE
    configs run): the fixture's few envs are tiled up to a whole number of workgroups."""
    meta, z = gu.load_fixture(name)
    be = EngineBackend(meta["E"], library=LIB, envs_per_workgroup=geom[0], threads_per_workgroup=geom[1], tile=tile,
                       **gu.ctor_kwargs(meta))
    assert be.env.engines[0].info.specialised == 1
    assert gu.replay(be, meta, z, steps=100 if tile <= 4 else 60) > 0   # (the GPU suite replays every trace in full)
    be.env.close()


def test_masked_reset_and_reseed():
    kw = rware_amd.env_kwargs("rware-tiny-2ag-v1")
    kw["reward_type"] = 1
    B = 6
    env = rware_amd.WarehouseVecEnv(B, library=LIB, envs_per_workgroup=4, threads_per_workgroup=64, **kw)
    orc = OracleVecEnv(B, **kw)
    env.reset(seed=5)
    orc.reset(seed=5)
    a = np.random.default_rng(0).integers(0, 5, size=(B, 2))
    for _ in range(5):
        env.step(a)
        orc.step_autoreset(a, "next_step")
    mask = np.array([1, 0, 0, 1, 0, 1], np.uint8)
    obs, _ = env.reset(mask=mask)                       # continue the streams of envs 0, 3, 5
    assert np.array_equal(obs, orc.reset(mask=mask))
    obs, _ = env.reset(seed=[11, 12, 13, 14, 15, 16], mask=mask)   # reseed just those
    for e in np.nonzero(mask)[0]:
        from rware_oracle import seed_state
        orc.rng[e] = seed_state(11 + int(e))
    assert np.array_equal(obs, orc.reset(mask=mask))
    st, so = env.get_state(), orc.get_state()
    for k in so:
        assert np.array_equal(st[k], so[k]), k
    env.close()


def test_host_layer_errors_and_views():
    kw = rware_amd.env_kwargs("rware-tiny-2ag-v1")
    env = rware_amd.WarehouseVecEnv(4, library=LIB, envs_per_workgroup=4, threads_per_workgroup=64, **kw)
    obs, info = env.reset(seed=0)
    assert obs.shape == (4, 2, 71) and obs.dtype == np.float32 and info == {}
    with pytest.raises(ValueError):
        env.step(np.full((4, 2), 5))
    with pytest.raises(AssertionError):
        env.step(np.zeros((4, 3), int))
    o, r, term, trunc, info = env.step([[rware_amd.Action.FORWARD, rware_amd.Action.NOOP]] * 4)
    assert r.shape == (4, 2) and term.dtype == bool and not trunc.any() and info == {}
    assert env.shelf_xy().shape == (4, env.n_shelves, 2)
    assert env.unwrapped is env and env.get_attr("n_agents") == (2, 2, 2, 2) and len(env.call("shelf_xy")) == 4
    with pytest.raises(NotImplementedError):
        env.render()
    ienv = rware_amd.WarehouseVecEnv(2, library=LIB, observation_type=rware_amd.ObservationType.IMAGE, **dict(kw, msg_bits=1))
    io, _ = ienv.reset(seed=0)                        # communication bits with image observations: actions (B, N, 2)
    assert io.shape == (2, 2, 5, 3, 3) and ienv.step(np.zeros((2, 2, 2), int))[0].shape == io.shape
    ienv.close()
    menv = rware_amd.WarehouseVecEnv(4, library=LIB, envs_per_workgroup=4, threads_per_workgroup=64, **dict(kw, msg_bits=2))
    mo, _ = menv.reset(seed=0)
    assert mo.shape == (4, 2, 8 + 9 * 9)              # L = 8 + (7 + M)(2r+1)^2
    with pytest.raises(ValueError):                   # message bits are binary
        menv.step(np.full((4, 2, 3), 2))
    with pytest.raises(AssertionError):               # [Action, bit, bit] per agent
        menv.step(np.zeros((4, 2), int))
    menv.close()
    denv = rware_amd.WarehouseVecEnv(2, library=LIB, observation_type=rware_amd.ObservationType.DICT, **kw)
    dobs, _ = denv.reset(seed=5)                      # DICT: the batched nested dict of :676-720
    flat, _ = rware_amd.WarehouseVecEnv(2, library=LIB, **kw).reset(seed=5)
    assert dobs["self"]["location"].dtype == np.int32 and np.array_equal(dobs["self"]["location"], flat[..., :2].astype(np.int32))
    assert len(dobs["sensors"]) == 9 and dobs["sensors"][4]["has_agent"].all()   # the centre cell is the agent itself
    assert list(dobs["sensors"][0]) == ["has_agent", "direction", "local_message", "has_shelf", "shelf_requested"]
    per_agent = denv.unbatch_dict_obs(dobs, 1)
    assert len(per_agent) == kw["n_agents"] and per_agent[0]["self"]["direction"] in (0, 1, 2, 3)
    denv.close()
    with pytest.raises(rware_amd._capi.EngineError):
        rware_amd.WarehouseVecEnv(2, library=LIB, envs_per_workgroup=6, **kw)   # not a multiple of 4
    env.close()


def test_sharded_env_equals_unsharded():
    """env-batch sharding (one engine per shard, global seeds) is invisible in the results."""
    kw = rware_amd.env_kwargs("rware-tiny-2ag-v1")
    one = rware_amd.WarehouseVecEnv(10, library=LIB, envs_per_workgroup=4, threads_per_workgroup=64, **kw)
    two = rware_amd.WarehouseVecEnv(10, library=LIB, devices=[0, 0, 0], envs_per_workgroup=4,
                                    threads_per_workgroup=64, **kw)
    assert len(two.engines) == 3
    o1, _ = one.reset(seed=123)
    o2, _ = two.reset(seed=123)
    assert np.array_equal(o1, o2)
    rng = np.random.default_rng(1)
    for _ in range(30):
        a = rng.integers(0, 5, size=(10, 2))
        r1, r2 = one.step(a), two.step(a)
        for x, y in zip(r1[:4], r2[:4]):
            assert np.array_equal(x, y)
    s1, s2 = one.get_state(), two.get_state()
    for k in s1:
        assert np.array_equal(s1[k], s2[k]), k
    one.close(); two.close()


@pytest.mark.parametrize("env_id,extra,B,geom,mode", [
    ("rware-small-4ag-v1", {"max_steps": 13}, 32, (0, 0), "next_step"),        # specialised kernel
    ("rware-tiny-2ag-v1", {"max_steps": 9}, 7, (4, 64), "same_step"),          # generic, ragged batch
    ("rware-medium-6ag-hard-v1", {"max_steps": 11, "reward_type": 0}, 16, (0, 0), "next_step"),
    ("rware-large-16ag-v1", {"sensor_range": 2, "max_steps": 10}, 5, (4, 128), "disabled"),
])
def test_fused_rollout_equals_stepwise_oracle(env_id, extra, B, geom, mode):
    """rw_step_many_device (one launch, env chunk resident in LDS across steps) == T single steps."""
    kw = rware_amd.env_kwargs(env_id)
    kw.update(extra)
    kw["reward_type"] = rware_amd.enums.enum_value(kw["reward_type"])
    env = rware_amd.WarehouseVecEnv(B, library=LIB, autoreset_mode=mode, envs_per_workgroup=geom[0],
                                    threads_per_workgroup=geom[1], **kw)
    orc = OracleVecEnv(B, **kw)
    assert np.array_equal(env.reset(seed=3)[0], orc.reset(seed=3))
    T = 37
    acts = np.random.default_rng(0).choice(5, size=(T, B, kw["n_agents"]), p=[.1, .55, .1, .1, .15]).astype(np.int32)
    obs, rew, term = env.rollout(acts)
    for t in range(T):
        o2, r2, d2 = orc.step_autoreset(acts[t], mode)
        assert np.array_equal(obs[t], o2) and np.array_equal(rew[t], r2) and np.array_equal(term[t], d2.astype(bool)), t
    st, so = env.get_state(), orc.get_state()
    for k in so:
        assert np.array_equal(st[k], so[k]), k
    # and the stepwise path continues from the fused one
    a = acts[0]
    o, r, d, _, _ = env.step(a)
    o2, r2, d2 = orc.step_autoreset(a, mode)
    assert np.array_equal(o, o2) and np.array_equal(r, r2)
    env.close()


@pytest.mark.parametrize("sensor_range", [4, 5])
def test_wide_sensor_ranges(sensor_range):
    """r = 4 (63-bit window rows) and r = 5 (77-bit rows, the multi-word path of the row gather)."""
    kw = rware_amd.env_kwargs("rware-tiny-3ag-v1")
    kw.update(sensor_range=sensor_range, max_steps=12)
    kw["reward_type"] = kw["reward_type"].value
    B = 5
    env = rware_amd.WarehouseVecEnv(B, library=LIB, envs_per_workgroup=4, threads_per_workgroup=64, **kw)
    orc = OracleVecEnv(B, **kw)
    assert env.obs_length == 8 + 7 * (2 * sensor_range + 1) ** 2
    assert np.array_equal(env.reset(seed=1)[0], orc.reset(seed=1))
    rng = np.random.default_rng(2)
    for t in range(30):
        a = rng.integers(0, 5, size=(B, 3))
        o, r, d, _, _ = env.step(a)
        o2, r2, d2 = orc.step_autoreset(a, "next_step")
        assert np.array_equal(o, o2) and np.array_equal(r, r2), t
    env.close()


@pytest.mark.parametrize("n_agents", [40, 64])
def test_many_agents_one_env_per_wavefront(n_agents):
    """N > 32: one env per wavefront in the agent phases (64 / N == 1), agent ids up to the 7-bit limit of the
    LDS agent layer, crowded enough that chains and cycles happen every step."""
    kw = rware_amd.env_kwargs("rware-medium-19ag-v1")
    kw.update(n_agents=n_agents, request_queue_size=n_agents, max_steps=25)
    kw["reward_type"] = kw["reward_type"].value
    B = 5
    env = rware_amd.WarehouseVecEnv(B, library=LIB, **kw)
    orc = OracleVecEnv(B, **kw)
    assert np.array_equal(env.reset(seed=3)[0], orc.reset(seed=3))
    rng = np.random.default_rng(8)
    for t in range(60):
        a = rng.choice(5, size=(B, n_agents), p=[.05, .65, .1, .1, .1])
        o, r, d, _, _ = env.step(a)
        o2, r2, d2 = orc.step_autoreset(a, "next_step")
        assert np.array_equal(o, o2) and np.array_equal(r, r2) and np.array_equal(d, d2.astype(bool)), t
    st, so = env.get_state(), orc.get_state()
    for k in so:
        assert np.array_equal(st[k], so[k]), k
    env.close()


def test_wide_shelf_ids_use_the_uint16_shadow():
    """More than 255 shelves (the reference's __main__ smoke layout, 29x28): shelf shadow is uint16."""
    kw = dict(shelf_columns=9, column_height=8, shelf_rows=3, n_agents=10, sensor_range=1, request_queue_size=5,
              max_inactivity_steps=None, max_steps=14, reward_type=0)
    B = 4
    env = rware_amd.WarehouseVecEnv(B, library=LIB, envs_per_workgroup=4, threads_per_workgroup=128, **kw)
    assert env.n_shelves > 255
    orc = OracleVecEnv(B, **kw)
    assert np.array_equal(env.reset(seed=8)[0], orc.reset(seed=8))
    rng = np.random.default_rng(5)
    for t in range(35):
        a = rng.choice(5, size=(B, 10), p=[.1, .5, .1, .1, .2])
        o, r, d, _, _ = env.step(a)
        o2, r2, d2 = orc.step_autoreset(a, "next_step")
        assert np.array_equal(o, o2) and np.array_equal(r, r2) and np.array_equal(d, d2.astype(bool)), t
    st, so = env.get_state(), orc.get_state()
    for k in so:
        assert np.array_equal(st[k], so[k]), k
    env.close()


def test_snapshot_restore_replays_bit_identically():
    kw = rware_amd.env_kwargs("rware-small-4ag-v1")
    kw["reward_type"] = 0
    kw["max_steps"] = 11
    B = 8
    env = rware_amd.WarehouseVecEnv(B, library=LIB, **kw)
    env.reset(seed=21)
    rng = np.random.default_rng(6)
    acts = rng.choice(5, size=(30, B, 4), p=[.1, .55, .1, .1, .15])
    for t in range(7):
        env.step(acts[t])
    snap = env.snapshot()
    saved = env.get_state()
    first = [env.step(acts[t]) for t in range(7, 30)]
    obs = env.restore(snap)
    back = env.get_state()
    for k in saved:
        assert np.array_equal(saved[k], back[k]), k
    second = [env.step(acts[t]) for t in range(7, 30)]
    for a, b in zip(first, second):
        for x, y in zip(a[:4], b[:4]):
            assert np.array_equal(x, y)
    env.free_snapshot(snap)
    env.close()


@pytest.mark.parametrize("obs_type,directional,layers,sr", [
    (2, True, None, 1),
    (2, False, [6, 0, 5], 2),          # ACCESSIBLE, SHELVES, GOALS in a custom channel order
    (3, True, [2, 1], 3),              # IMAGE_DICT with two layers
])
def test_image_observations_match_oracle(obs_type, directional, layers, sr):
    kw = rware_amd.env_kwargs("rware-small-5ag-v1")
    kw.update(sensor_range=sr, max_steps=16)
    kw["reward_type"] = kw["reward_type"].value
    extra = dict(observation_type=obs_type, image_observation_directional=directional, image_observation_layers=layers)
    B = 6
    env = rware_amd.WarehouseVecEnv(B, library=LIB, envs_per_workgroup=4, threads_per_workgroup=128, **kw, **extra)
    orc = OracleVecEnv(B, **kw, **extra)

    def same(a, b):
        if isinstance(a, dict):
            return np.array_equal(a["image"], b[0]) and np.array_equal(a["features"], b[1])
        return np.array_equal(a, b)

    assert same(env.reset(seed=4)[0], orc.reset(seed=4))
    rng = np.random.default_rng(7)
    acts = rng.choice(5, size=(40, B, 5), p=[.1, .5, .15, .15, .1])
    for t in range(25):
        o, r, d, _, _ = env.step(acts[t])
        o2, r2, d2 = orc.step_autoreset(acts[t], "next_step")
        assert same(o, o2) and np.array_equal(r, r2), t
    img, rew, term = env.rollout(acts[25:])      # fused rollout writes the image tape
    for t in range(25, 40):
        o2, r2, d2 = orc.step_autoreset(acts[t], "next_step")
        assert np.array_equal(img[t - 25], o2[0] if isinstance(o2, tuple) else o2) and np.array_equal(rew[t - 25], r2), t
    env.close()


SQUARE = dict(shelf_columns=3, column_height=3, shelf_rows=2, n_agents=5, msg_bits=0, sensor_range=2,
              request_queue_size=3, max_inactivity_steps=None, max_steps=30, reward_type=1)   # a 10 x 10 grid


@pytest.mark.parametrize("obs_type,directional,layers", [
    (2, True, [3, 4, 0, 2]),           # AGENT_DIRECTION, AGENT_LOAD, SHELVES, AGENTS
    (3, False, [4, 5, 3]),             # IMAGE_DICT, north-up: AGENT_LOAD, GOALS, AGENT_DIRECTION
])
def test_transposed_image_layers_match_oracle(obs_type, directional, layers):
    """AGENT_DIRECTION / AGENT_LOAD as the reference writes them, layer[ag.x, ag.y] (:552, :558); on a square
    grid the transposed index is always in bounds."""
    extra = dict(observation_type=obs_type, image_observation_directional=directional, image_observation_layers=layers)
    B = 6
    env = rware_amd.WarehouseVecEnv(B, library=LIB, envs_per_workgroup=4, threads_per_workgroup=128, **SQUARE, **extra)
    assert tuple(env.grid_size) == (10, 10)
    orc = OracleVecEnv(B, **SQUARE, **extra)

    def same(a, b):
        if isinstance(a, dict):
            return np.array_equal(a["image"], b[0]) and np.array_equal(a["features"], b[1])
        return np.array_equal(a, b)

    assert same(env.reset(seed=11)[0], orc.reset(seed=11))
    rng = np.random.default_rng(3)
    acts = rng.choice(5, size=(70, B, 5), p=[.1, .45, .15, .15, .15])
    seen_dir = seen_load = 0
    for t in range(50):
        o, r, d, _, _ = env.step(acts[t])
        o2, r2, d2 = orc.step_autoreset(acts[t], "next_step")
        assert same(o, o2) and np.array_equal(r, r2), t
        img = o["image"] if isinstance(o, dict) else o
        seen_dir += int((img[:, :, layers.index(3)] > 1).sum())
        seen_load += int(img[:, :, layers.index(4)].sum())
    assert seen_dir > 0 and seen_load > 0          # the layers were exercised (values 2..4, loaded agents in view)
    img, rew, term = env.rollout(acts[50:])
    for t in range(50, 70):
        o2, r2, d2 = orc.step_autoreset(acts[t], "next_step")
        assert np.array_equal(img[t - 50], o2[0] if isinstance(o2, tuple) else o2) and np.array_equal(rew[t - 50], r2), t
    env.close()


@pytest.mark.parametrize("layer", [3, 4])
def test_transposed_image_layers_raise_indexerror_like_the_reference(layer):
    """On every registered layout H > W, so the reference's layer[ag.x, ag.y] raises IndexError once an agent
    (a loaded one for AGENT_LOAD) reaches y >= W; the engine reports it at the same step."""
    kw = rware_amd.env_kwargs("rware-tiny-2ag-v1")
    kw["reward_type"] = kw["reward_type"].value
    extra = dict(observation_type=2, image_observation_layers=[2, layer])
    B = 4
    env = rware_amd.WarehouseVecEnv(B, library=LIB, envs_per_workgroup=4, threads_per_workgroup=64, **kw, **extra)
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
    while not e1 and t < 400:
        a = rng.choice(5, size=(B, 2), p=[.05, .5, .15, .15, .15])
        (res, e1), (res2, e2) = attempt(lambda: env.step(a)), attempt(lambda: orc.step_autoreset(a, "next_step"))
        assert e1 == e2, t
        if not e1:
            assert np.array_equal(res[0], res2[0]), t
        t += 1
    assert e1, "no agent ever reached y >= W"
    env.close()


def test_communication_bits_rollout_and_state():
    kw = rware_amd.env_kwargs("rware-small-6ag-v1")
    kw.update(msg_bits=2, max_steps=14)
    kw["reward_type"] = kw["reward_type"].value
    B = 5
    env = rware_amd.WarehouseVecEnv(B, library=LIB, envs_per_workgroup=4, threads_per_workgroup=128, autoreset_mode="same_step", **kw)
    orc = OracleVecEnv(B, **kw)
    assert np.array_equal(env.reset(seed=2)[0], orc.reset(seed=2))
    rng = np.random.default_rng(8)
    acts = np.concatenate([rng.choice(5, size=(40, B, 6, 1), p=[.1, .5, .15, .15, .1]), rng.integers(0, 2, size=(40, B, 6, 2))], axis=-1)
    for t in range(15):
        o, r, d, _, _ = env.step(acts[t])
        o2, r2, d2 = orc.step_autoreset(acts[t], "same_step")
        assert np.array_equal(o, o2) and np.array_equal(r, r2) and np.array_equal(d, d2.astype(bool)), t
    obs, rew, term = env.rollout(acts[15:])
    for t in range(15, 40):
        o2, r2, d2 = orc.step_autoreset(acts[t], "same_step")
        assert np.array_equal(obs[t - 15], o2) and np.array_equal(rew[t - 15], r2), t
    st, so = env.get_state(), orc.get_state()
    for k in so:
        assert np.array_equal(st[k], so[k]), k
    env.close()


def test_step_tape_device_equals_stepwise_calls():
    """rw_step_tape_device: n per-step launches from a device action tape (wrapping around) == the same steps submitted
    one host call at a time."""
    import ctypes as C
    kw = rware_amd.env_kwargs("rware-tiny-2ag-v1")
    kw["max_steps"] = 9
    B, TS = 6, 5
    a = rware_amd.WarehouseVecEnv(B, library=LIB, envs_per_workgroup=4, threads_per_workgroup=64, **kw)
    b = rware_amd.WarehouseVecEnv(B, library=LIB, envs_per_workgroup=4, threads_per_workgroup=64, **kw)
    a.reset(seed=11)
    b.reset(seed=11)
    tape = np.random.default_rng(0).integers(0, 5, size=(TS, B, 2), dtype=np.int32)
    eng = a.engines[0]
    d = C.c_void_p()
    eng._check(eng.lib.rw_device_malloc(eng._h, tape.nbytes, C.byref(d)))
    eng._check(eng.lib.rw_copy_to_device(eng._h, d, tape.ctypes.data, tape.nbytes))
    eng.step_tape_device(d.value, TS, 3, 7)           # rows 3, 4, 0, 1, 2, 3, ...
    eng.step_tape_device_timed(d.value, TS, (3 + 7) % TS, 5, 0, 1)   # same launches, timing events on the first / last one
    eng.sync()
    assert eng.event_elapsed_ms(0, 1) >= 0.0
    with pytest.raises(Exception):
        eng.step_tape_device_timed(d.value, TS, 0, 0, 0, 1)   # nothing to attach the events to
    with pytest.raises(Exception):
        eng.step_tape_device_timed(d.value, TS, 0, 1, 2, 2)   # one slot for both ends
    for k in range(12):
        obs, rew, term, _, _ = b.step(tape[(3 + k) % TS])
    assert np.array_equal(a.observations(), obs)
    assert np.array_equal(a.engines[0].read("rewards"), rew)
    sa, sb = a.get_state(), b.get_state()
    for k in sa:
        assert np.array_equal(sa[k], sb[k]), k
    eng.lib.rw_device_free(eng._h, d)
    a.close(); b.close()


def test_grid_is_a_derived_view_refreshed_on_demand():
    """RW_BUF_GRID is rebuilt from the shelf layer and the agent coordinates when asked for: get_state() / a fresh
    rw_get_buffer are current; a pointer borrowed earlier lags behind later steps until rw_refresh_grid."""
    import ctypes as C
    kw = rware_amd.env_kwargs("rware-tiny-2ag-v1")
    kw["reward_type"] = 1
    B = 4
    env = rware_amd.WarehouseVecEnv(B, library=LIB, envs_per_workgroup=4, threads_per_workgroup=64, **kw)
    orc = OracleVecEnv(B, **kw)
    env.reset(seed=3)
    orc.reset(seed=3)
    da = env.engines[0].device_array("grid")                      # borrowed pointer (host memory in the emulation)
    view = np.ctypeslib.as_array((C.c_int32 * int(np.prod(da.shape))).from_address(da.ptr)).reshape(da.shape)
    assert np.array_equal(view, orc.get_state()["grid"])          # current when handed out
    rng = np.random.default_rng(0)
    before = view.copy()
    for t in range(12):
        a = rng.choice(5, size=(B, 2), p=[0.1, 0.6, 0.1, 0.1, 0.1]).astype(np.int32)
        env.step(a)
        orc.step_autoreset(a, "next_step")
    want = orc.get_state()["grid"]
    assert not np.array_equal(want, before), "nothing moved: the test needs a different seed"
    assert np.array_equal(view, before)                            # the steps did not touch the exported grid
    env.refresh_grid()
    assert np.array_equal(view, want)                              # ... rw_refresh_grid brings it up to date
    a = rng.integers(0, 5, size=(B, 2), dtype=np.int32)
    env.step(a)
    orc.step_autoreset(a, "next_step")
    assert np.array_equal(env.get_state()["grid"], orc.get_state()["grid"])   # get_state() is always current
    env.close()


# --------------------------------------------------------------------------- round-3 additions (advisor findings)
@pytest.mark.parametrize("name,geom,tile", [
    ("tiny-2ag", (0, 0), 4),           # pair exchange (N = 2)
    ("small-4ag", (0, 0), 2),          # quad exchange (N = 4)
    ("medium-6ag-hard", (0, 0), 8),    # ds_bpermute exchange (N = 6)
])
def test_emulated_exact_shape_builds_full_length_golden(name, geom, tile):
    """One FULL-length golden replay per register-exchange flavour on the emulation: the late part of a trace is where the
    deliveries, the request replacement (queue written back only when dirty), termination and the autoreset are."""
    meta, z = gu.load_fixture(name)
    be = EngineBackend(meta["E"], library=LIB, envs_per_workgroup=geom[0], threads_per_workgroup=geom[1], tile=tile,
                       **gu.ctor_kwargs(meta))
    assert be.env.engines[0].info.specialised == 1
    # (the N = 6 trace is 650 steps x 24 tiled envs of 6 agents — a minute on host threads: its first 250 steps, which hold the
    #  first deliveries and queue replacements; the other two flavours in full.  The GPU suite replays every trace in full.)
    n = gu.replay(be, meta, z, steps=250 if name == "medium-6ag-hard" else None)
    assert n >= 250
    be.env.close()


def test_layouts_wider_than_256_cells_keep_exact_coordinates():
    """x, y travel as bytes through the single-pass observation expansion only when the grid fits 256 x 256; a 4 x 268 grid
    (shelf_columns=89, column_height=1) must take the float path: agents at x >= 256 report x, not x mod 256."""
    kw = dict(shelf_columns=89, column_height=1, shelf_rows=1, n_agents=8, msg_bits=0, sensor_range=1, request_queue_size=4,
              max_inactivity_steps=None, max_steps=500, reward_type=1)
    B = 6
    env = rware_amd.WarehouseVecEnv(B, library=LIB, envs_per_workgroup=4, threads_per_workgroup=64, **kw)
    assert env.grid_size[1] > 256
    orc = OracleVecEnv(B, **kw)
    obs, _ = env.reset(seed=21)
    assert np.array_equal(obs, orc.reset(seed=21))
    assert (obs[..., 0] >= 256).any(), "no agent beyond x = 255: pick another seed"
    rng = np.random.default_rng(2)
    for t in range(8):
        a = rng.choice(5, size=(B, 8), p=[0.1, 0.6, 0.1, 0.1, 0.1]).astype(np.int32)
        obs, rew, term, _, _ = env.step(a)
        o2, r2, d2 = orc.step_autoreset(a, "next_step")
        assert np.array_equal(obs, o2), t
    env.close()


def test_coordinate_writes_mark_the_derived_grid_stale_and_truncated_is_read_only():
    kw = rware_amd.env_kwargs("rware-tiny-2ag-v1")
    env = rware_amd.WarehouseVecEnv(4, library=LIB, envs_per_workgroup=4, threads_per_workgroup=64, **kw)
    env.reset(seed=1)
    st = env.get_state()
    ax, ay = st["agent_x"].copy(), st["agent_y"].copy()
    free = (st["grid"][:, 0] == 0) & (st["grid"][:, 1] == 0)
    for e in range(4):                                    # teleport agent 0 of every env to a free cell
        y, x = np.argwhere(free[e])[0]
        ax[e, 0], ay[e, 0] = x, y
    env.set_state(agent_x=ax, agent_y=ay)
    g = env.get_state()["grid"]
    for e in range(4):
        assert g[e, 0, ay[e, 0], ax[e, 0]] == 1            # layer 0 follows the coordinates right away, not after a step
        assert (g[e, 0] == 1).sum() == 1
    # grid and coordinates given together, in either keyword order: the caller's coordinates win for layer 0
    env.set_state(agent_y=st["agent_y"], grid=st["grid"], agent_x=st["agent_x"])
    assert np.array_equal(env.get_state()["grid"], st["grid"])
    with pytest.raises(Exception):
        env.engines[0].write("truncated", np.ones(4, np.uint8))
    assert not env.engines[0].read("truncated").any()
    env.close()


@pytest.mark.parametrize("env_id", [
    "rware-tiny-8ag-v1", "rware-tiny-6ag-easy-v1", "rware-small-2ag-hard-v1", "rware-small-8ag-easy-v1",
    "rware-medium-4ag-v1", "rware-medium-8ag-hard-v1",
])
def test_paper_task_grid_runs_exact_shape_builds(env_id):
    """tiny / small / medium x 2, 4, 6, 8 agents x easy / normal / hard each have an exact-shape build (rware_static_table.h):
    a sample of them against the oracle across an autoreset (N = 8: exact-shape build with the LDS exchange)."""
    kw = rware_amd.env_kwargs(env_id)
    kw["max_steps"] = 18
    kw["reward_type"] = rware_amd.enums.enum_value(kw["reward_type"])
    B = 16
    env = rware_amd.WarehouseVecEnv(B, library=LIB, **kw)
    # (6 and 8 agents: the half-size-workgroup build serves batches up to 16384 envs)
    assert env.engines[0].info.specialised == 1 and env.engines[0].info.envs_per_workgroup == (8 if kw["n_agents"] >= 6 else 16)
    orc = OracleVecEnv(B, **kw)
    assert np.array_equal(env.reset(seed=4)[0], orc.reset(seed=4))
    rng = np.random.default_rng(6)
    for t in range(30):
        a = rng.choice(5, size=(B, kw["n_agents"]), p=[.1, .55, .1, .1, .15]).astype(np.int32)
        obs, rew, term, _, _ = env.step(a)
        o2, r2, d2 = orc.step_autoreset(a, "next_step")
        assert np.array_equal(obs, o2) and np.array_equal(rew, r2) and np.array_equal(term, d2.astype(bool)), t
    st, so = env.get_state(), orc.get_state()
    for k in so:
        assert np.array_equal(st[k], so[k]), k
    env.close()


def test_agent_arrays_are_derived_views_of_the_packed_records():
    """RW_BUF_AGENT_X .. _DELIVERED are unpacked from the packed agent records when asked for: a pointer borrowed earlier lags
    behind later steps until rw_get_buffer is called again (same pointer, refreshed); a host write of ONE view keeps the
    other four (the records are re-packed from all five)."""
    import ctypes as C
    kw = rware_amd.env_kwargs("rware-tiny-2ag-v1")
    kw["reward_type"] = 1
    B = 4
    env = rware_amd.WarehouseVecEnv(B, library=LIB, envs_per_workgroup=4, threads_per_workgroup=64, **kw)
    orc = OracleVecEnv(B, **kw)
    env.reset(seed=3)
    orc.reset(seed=3)
    eng = env.engines[0]
    da = eng.device_array("agent_x")                               # borrowed pointer (host memory in the emulation)
    view = np.ctypeslib.as_array((C.c_int32 * int(np.prod(da.shape))).from_address(da.ptr)).reshape(da.shape)
    assert np.array_equal(view, orc.get_state()["agent_x"])
    rng = np.random.default_rng(0)
    before = view.copy()
    for t in range(12):
        a = rng.choice(5, size=(B, 2), p=[0.1, 0.6, 0.1, 0.1, 0.1]).astype(np.int32)
        env.step(a)
        orc.step_autoreset(a, "next_step")
    so = orc.get_state()
    assert not np.array_equal(so["agent_x"], before), "nothing moved: the test needs a different seed"
    assert np.array_equal(view, before)                            # the steps did not touch the exported array
    assert eng.device_array("agent_x").ptr == da.ptr               # same buffer ...
    assert np.array_equal(view, so["agent_x"])                     # ... now current
    # write one view: the other four survive, and the engine continues from the injected state
    st = env.get_state()
    new_dir = (st["agent_dir"] + 1) % 4
    env.set_state(agent_dir=new_dir)
    orc.set_state(agent_dir=new_dir)
    st2 = env.get_state()
    for k in ("agent_x", "agent_y", "agent_carry", "agent_delivered"):
        assert np.array_equal(st2[k], st[k]), k
    assert np.array_equal(st2["agent_dir"], new_dir)
    for t in range(6):
        a = rng.integers(0, 5, size=(B, 2), dtype=np.int32)
        obs, _, _, _, _ = env.step(a)
        o2, _, _ = orc.step_autoreset(a, "next_step")
        assert np.array_equal(obs, o2), t
    env.close()


def test_two_agent_tasks_take_double_size_workgroups_from_16384_envs():
    """2 agents: the 32-env build (64 agents = one full agent wavefront) from 16384 envs up, the 16-env build below and for
    batches that are no multiple of 32; pinned here through the explicit geometry and checked against the oracle."""
    kw = rware_amd.env_kwargs("rware-small-2ag-hard-v1")
    kw["max_steps"] = 15
    kw["reward_type"] = rware_amd.enums.enum_value(kw["reward_type"])
    B = 64
    env = rware_amd.WarehouseVecEnv(B, library=LIB, envs_per_workgroup=32, threads_per_workgroup=256, **kw)
    assert env.engines[0].info.specialised == 1 and env.engines[0].info.envs_per_workgroup == 32
    small = rware_amd.WarehouseVecEnv(B, library=LIB, **kw)
    assert small.engines[0].info.envs_per_workgroup == 16          # default geometry below 16384 envs
    small.close()
    orc = OracleVecEnv(B, **kw)
    assert np.array_equal(env.reset(seed=8)[0], orc.reset(seed=8))
    rng = np.random.default_rng(1)
    for t in range(24):
        a = rng.choice(5, size=(B, 2), p=[.1, .55, .1, .1, .15]).astype(np.int32)
        obs, rew, term, _, _ = env.step(a)
        o2, r2, d2 = orc.step_autoreset(a, "next_step")
        assert np.array_equal(obs, o2) and np.array_equal(rew, r2) and np.array_equal(term, d2.astype(bool)), t
    acts = rng.choice(5, size=(10, B, 2), p=[.1, .55, .1, .1, .15]).astype(np.int32)
    obs, rew, term = env.rollout(acts)
    for k in range(10):
        o2, r2, d2 = orc.step_autoreset(acts[k], "next_step")
        assert np.array_equal(obs[k], o2) and np.array_equal(rew[k], r2), k
    env.close()


def test_seed_and_global_image_methods():
    """The rest of Warehouse's public surface: seed() (:962-964) re-seeds the streams without resetting; get_global_image()
    (:966-1040) with the reference's cache semantics."""
    from rware_oracle import seed_state
    kw = rware_amd.env_kwargs("rware-tiny-2ag-v1")
    env = rware_amd.WarehouseVecEnv(4, library=LIB, envs_per_workgroup=4, threads_per_workgroup=64, **kw)
    env.reset(seed=1)
    before = env.get_state()
    env.seed(77)
    after = env.get_state()
    for i in range(4):
        assert np.array_equal(after["rng"][i], seed_state(77 + i))
    for k in before:
        if k != "rng":
            assert np.array_equal(before[k], after[k]), k       # nothing but the streams changed
    env.seed(None)
    assert np.array_equal(env.get_state()["rng"], after["rng"])
    img = env.get_global_image()
    assert img.shape == (4, 2, 11, 10) and img.dtype == np.float32
    assert np.array_equal(img[:, 0] > 0, after["grid"][:, 1] > 0) and img[:, 1].sum() == 2 * 4     # SHELVES, GOALS (2 goal cells)
    env.step(np.full((4, 2), 1))
    assert env.get_global_image([2]) is img                      # cached, like the reference (recompute=False)
    img2 = env.get_global_image([2], recompute=True)
    assert img2.shape == (4, 1, 11, 10) and np.array_equal(img2[:, 0] > 0, env.get_state()["grid"][:, 0] > 0)
    env.close()


@pytest.mark.parametrize("env_id,extra", [
    ("rware-small-3ag-v1", {}), ("rware-small-5ag-easy-v1", {}), ("rware-small-7ag-hard-v1", {}), ("rware-small-1ag-hard-v1", {}),
    ("rware-small-1ag-easy-v1", {}), ("rware-small-4ag-v1", {"request_queue_size": 5}), ("rware-small-6ag-v1", {"request_queue_size": 1}),
    ("rware-small-8ag-v1", {"request_queue_size": 11}), ("rware-small-2ag-v1", {"request_queue_size": 3}),
    ("rware-tiny-3ag-v1", {}), ("rware-medium-5ag-easy-v1", {}), ("rware-large-7ag-v1", {}), ("rware-large-2ag-hard-v1", {}),
    ("rware-large-4ag-v1", {}),
    # 9 .. 19 agents (round 4): agent phases in registers for every registered agent count — ds_bpermute exchange, chain links
    # in 64 bits (N <= 12) or 128 bits (N >= 13), two or three agent wavefronts per 8-env workgroup
    ("rware-small-9ag-hard-v1", {}), ("rware-small-10ag-v1", {}), ("rware-small-12ag-easy-v1", {}), ("rware-tiny-13ag-v1", {}),
    ("rware-medium-14ag-v1", {}), ("rware-medium-16ag-hard-v1", {}), ("rware-large-17ag-v1", {}), ("rware-small-19ag-v1", {}),
    ("rware-large-16ag-v1", {"request_queue_size": 7}), ("rware-tiny-15ag-easy-v1", {}),
])
def test_agent_count_static_builds_read_the_queue_length_at_run_time(env_id, extra):
    """Tasks without an exact (N, Q) entry run the agent-count-static build of their size and agent count (Q == -1 in
    rware_static_table.h: any queue length up to 2 N, the LDS carve-up reserves 2 N slots): against the oracle across an
    autoreset, per-step launches and a fused rollout; Q = 0 (1 agent, hard) included."""
    kw = rware_amd.env_kwargs(env_id)
    kw.update(extra)
    kw["max_steps"] = 18
    kw["reward_type"] = rware_amd.enums.enum_value(kw["reward_type"])
    B, N = 32, kw["n_agents"]
    env = rware_amd.WarehouseVecEnv(B, library=LIB, **kw)
    assert env.engines[0].info.specialised == 1
    # (geometry by the measured rules of rware_static_table.h / rw_create: 16 envs up to 4 agents, 8 from 5 on — except the per-step
    #  launches of 13 .. 16 agents below one round of 8-env workgroups, which run on the 4-env build where the warehouse size has one;
    #  the fused rollout further down then runs on the 8-env build of the same engine)
    wide4 = 13 <= N <= 16 and "tiny" not in env_id
    assert env.engines[0].info.build_kind == 2 and env.engines[0].info.envs_per_workgroup == (16 if N <= 4 else 4 if wide4 else 8)
    orc = OracleVecEnv(B, **kw)
    assert np.array_equal(env.reset(seed=14)[0], orc.reset(seed=14))
    rng = np.random.default_rng(16)
    for t in range(26):
        a = rng.choice(5, size=(B, N), p=[.1, .5, .1, .1, .2]).astype(np.int32)
        obs, rew, term, _, _ = env.step(a)
        o2, r2, d2 = orc.step_autoreset(a, "next_step")
        assert np.array_equal(obs, o2) and np.array_equal(rew, r2) and np.array_equal(term, d2.astype(bool)), t
    acts = rng.choice(5, size=(12, B, N), p=[.1, .5, .1, .1, .2]).astype(np.int32)
    obs, rew, term = env.rollout(acts)
    for k in range(12):
        o2, r2, d2 = orc.step_autoreset(acts[k], "next_step")
        assert np.array_equal(obs[k], o2) and np.array_equal(rew[k], r2) and np.array_equal(term[k], d2.astype(bool)), k
    st, so = env.get_state(), orc.get_state()
    for k in so:
        assert np.array_equal(st[k], so[k]), k
    env.close()


@pytest.mark.parametrize("name,tile", [("small-4ag", 4), ("small-8ag-global-inact", 16)])
def test_agent_count_static_builds_replay_reference_golden(monkeypatch, name, tile):
    """The reference's golden traces on the agent-count-static builds (RWARE_PREFER_QRT=1 makes rw_create skip the exact
    (N, Q) entries of the task grid): deliveries, queue replacement and termination with the queue length a run-time value."""
    monkeypatch.setenv("RWARE_PREFER_QRT", "1")
    meta, z = gu.load_fixture(name)
    kw = gu.ctor_kwargs(meta)
    if name == "small-4ag":
        kw = dict(kw)
    be = EngineBackend(meta["E"], library=LIB, tile=tile, **kw)
    assert be.env.engines[0].info.build_kind == 2
    assert gu.replay(be, meta, z, steps=60 if tile > 4 else 150) > 0
    be.env.close()


def test_observation_store_policy_is_a_create_time_choice():
    """rw_stream_flags RW_OBS_STORES_CACHED / _STREAM (and the engine's own rule when neither is given) only change HOW the
    observation lines are stored; rw_info reports the choice, the results are the same."""
    kw = rware_amd.env_kwargs("rware-small-4ag-v1")
    envs = {pol: rware_amd.WarehouseVecEnv(16, library=LIB, obs_stores=pol, **kw) for pol in (None, "cached", "stream")}
    assert envs[None].engines[0].info.obs_stores_stream == 1       # service-wave workgroups: the hint by default
    assert envs["cached"].engines[0].info.obs_stores_stream == 0 and envs["stream"].engines[0].info.obs_stores_stream == 1
    big = rware_amd.WarehouseVecEnv(8, library=LIB, **dict(rware_amd.env_kwargs("rware-large-16ag-v1"), sensor_range=2))
    assert big.engines[0].info.obs_stores_stream == 0              # a large observation chunk per workgroup, below the cache size
    big.close()
    obs = {pol: e.reset(seed=2)[0] for pol, e in envs.items()}
    rng = np.random.default_rng(0)
    for t in range(6):
        a = rng.integers(0, 5, size=(16, 4), dtype=np.int32)
        obs = {pol: e.step(a)[0] for pol, e in envs.items()}
        assert np.array_equal(obs[None], obs["cached"]) and np.array_equal(obs[None], obs["stream"])
    for e in envs.values():
        e.close()


@pytest.mark.parametrize("case", range(12))
def test_generic_kernel_matches_oracle_on_random_shapes(case):
    """The generic (every shape at run time) kernel on 12 random warehouses — rows / columns / column height, 1..10 agents,
    queue length, sensor range 1..4, reward type, inactivity limit, normalised coordinates, ragged batch — against the oracle,
    per-step launches across autoresets and a fused rollout."""
    g = np.random.default_rng(7000 + case)
    rows, cols, height = int(g.integers(1, 4)), int(g.choice([3, 5])), int(g.integers(1, 9))
    n_agents = int(g.integers(1, 11))
    kw = dict(shelf_columns=cols, column_height=height, shelf_rows=rows, n_agents=n_agents, msg_bits=0,
              sensor_range=int(g.integers(1, 5)), request_queue_size=int(g.integers(0, min(2 * n_agents, rows * cols * height // 2) + 1)),
              max_inactivity_steps=(None if g.random() < 0.6 else int(g.integers(8, 25))), max_steps=int(g.integers(12, 30)),
              reward_type=int(g.integers(0, 3)), normalised_coordinates=bool(g.random() < 0.25))
    B = int(g.integers(3, 10))
    mode = ["next_step", "same_step", "disabled"][case % 3]
    env = rware_amd.WarehouseVecEnv(B, library=LIB, envs_per_workgroup=4, threads_per_workgroup=int(g.choice([64, 128])), autoreset_mode=mode, **kw)
    orc = OracleVecEnv(B, **kw)
    assert np.array_equal(env.reset(seed=70 + case)[0], orc.reset(seed=70 + case))
    rng = np.random.default_rng(case)
    for t in range(40):
        a = rng.choice(5, size=(B, n_agents), p=[.1, .55, .1, .1, .15]).astype(np.int32)
        obs, rew, term, _, _ = env.step(a)
        o2, r2, d2 = orc.step_autoreset(a, mode)
        assert np.array_equal(obs, o2) and np.array_equal(rew, r2) and np.array_equal(term, d2.astype(bool)), (t, kw)
        if mode == "disabled" and term.any():
            m = term.astype(np.uint8)
            assert np.array_equal(env.reset(mask=m)[0], orc.reset(mask=m)), t
    acts = rng.choice(5, size=(8, B, n_agents), p=[.1, .55, .1, .1, .15]).astype(np.int32)
    if mode != "disabled":
        obs, rew, term = env.rollout(acts)
        for k in range(8):
            o2, r2, d2 = orc.step_autoreset(acts[k], mode)
            assert np.array_equal(obs[k], o2) and np.array_equal(rew[k], r2), (k, kw)
    st, so = env.get_state(), orc.get_state()
    for k in so:
        assert np.array_equal(st[k], so[k]), (k, kw)
    env.close()


@pytest.mark.parametrize("env_id,p_forward", [
    ("rware-tiny-13ag-v1", 0.75), ("rware-tiny-16ag-v1", 0.75), ("rware-tiny-17ag-hard-v1", 0.7), ("rware-tiny-19ag-v1", 0.8),
    ("rware-tiny-10ag-v1", 0.8), ("rware-tiny-12ag-easy-v1", 0.75),
])
def test_crowded_warehouses_resolve_long_chains(env_id, p_forward):
    """9 .. 19 agents on the 110 cells of the tiny warehouse under a forward-heavy policy: long follower chains, contested
    cells with unequal depths, blocked tails and cycles on nearly every step — the per-cell agent phases of the per-step kernel
    (occupant from the agent layer, winners among the target's four neighbours, pointer-chased depth and chain walk) and the
    register-exchange ones of the fused rollout (wide priority words, 128-bit chain links) against the oracle's literal networkx
    restatement."""
    kw = rware_amd.env_kwargs(env_id)
    kw["max_steps"] = 40
    kw["reward_type"] = rware_amd.enums.enum_value(kw["reward_type"])
    B, N = 16, kw["n_agents"]
    env = rware_amd.WarehouseVecEnv(B, library=LIB, **kw)
    assert env.engines[0].info.build_kind == 2 and env.engines[0].info.envs_per_workgroup == 8    # (the tiny warehouse: 8-env builds only)
    orc = OracleVecEnv(B, **kw)
    assert np.array_equal(env.reset(seed=31)[0], orc.reset(seed=31))
    rng = np.random.default_rng(33)
    rest = (1.0 - p_forward) / 4
    for t in range(90):
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


@pytest.mark.parametrize("threads", ["1", "0"])
def test_rw_multi_enqueues_every_engine_from_one_call(threads, monkeypatch):
    """rw_multi ("a single C call that fans out"): four engines — the shards of a single-process multi-device
    env — stepped through ONE rw_multi_step_device call per round (launcher thread per engine), against the same four shards
    stepped one rw_step_device call each: identical state, and the engines stay usable on their own afterwards."""
    from rware_amd import _capi
    monkeypatch.setenv("RWARE_MULTI_THREADS", threads)   # both modes: a launcher thread per engine / one loop in the caller's thread
    kw = rware_amd.env_kwargs("rware-tiny-2ag-v1")
    mk = lambda: rware_amd.WarehouseVecEnv(32, library=LIB, devices=[0, 0, 0, 0], max_steps=12, **{k: v for k, v in kw.items() if k != "max_steps"})
    a, b = mk(), mk()
    assert len(a.engines) == 4
    a.reset(seed=5); b.reset(seed=5)
    multi = _capi.MultiEngine(a.engines)
    rng = np.random.default_rng(1)
    bufs = [np.zeros((8, 2), np.int32) for _ in range(4)]     # (the emulation build's "device" memory is host memory)
    for t in range(30):
        acts = rng.integers(0, 5, size=(32, 2), dtype=np.int32)
        for k in range(4):
            bufs[k][:] = acts[8 * k:8 * k + 8]
        multi.step_device([x.ctypes.data for x in bufs])
        for k, eng in enumerate(b.engines):
            eng.step_device(bufs[k].ctypes.data)
    sa, sb = a.get_state(), b.get_state()
    for k in sa:
        assert np.array_equal(sa[k], sb[k]), k
    assert np.array_equal(a.observations(), b.observations()) and sa["steps"].max() <= 12
    a.engines[2].step_device(bufs[2].ctypes.data)             # still an ordinary engine
    with pytest.raises(_capi.EngineError):
        multi.step_device([bufs[0].ctypes.data, 0, bufs[2].ctypes.data, bufs[3].ctypes.data])   # a NULL action pointer
    multi.close(); a.close(); b.close()


def test_start_stagger_is_a_create_time_rule_and_does_not_change_results(monkeypatch):
    """Launches of two or more rounds of workgroups (>= 2 x 8 per CU; the emulated device has one CU) stagger the start of the first
    eight workgroups per CU (rw_info.stagger_ticks x 10 ns per slot; RWARE_STAGGER_TICKS moves the default) — where they do not run at
    raised wavefront priority (round 6: with the priority the delay is only a delay): a delay, never a different result."""
    kw = rware_amd.env_kwargs("rware-small-4ag-v1")
    kw["max_steps"] = 11
    kw["reward_type"] = rware_amd.enums.enum_value(kw["reward_type"])
    prio = rware_amd.WarehouseVecEnv(256, library=LIB, **kw)             # 16 workgroups, two rounds — at raised priority: no stagger
    assert (prio.engines[0].info.wave_priority & 1, prio.engines[0].info.stagger_ticks) == (1, 0)
    prio.close()
    monkeypatch.setenv("RWARE_PRIO", "0")                                # the rest of this test: the rule without the priority
    small = rware_amd.WarehouseVecEnv(64, library=LIB, **kw)             # 4 workgroups of 16 envs
    assert small.engines[0].info.stagger_ticks == 0
    small.close()
    monkeypatch.setenv("RWARE_STAGGER_TICKS", "0")
    off = rware_amd.WarehouseVecEnv(256, library=LIB, **kw)
    assert off.engines[0].info.stagger_ticks == 0
    off.close()
    monkeypatch.setenv("RWARE_STAGGER_TICKS", "40")
    forced = rware_amd.WarehouseVecEnv(64, library=LIB, **kw)
    assert forced.engines[0].info.stagger_ticks == 40
    forced.close()
    monkeypatch.delenv("RWARE_STAGGER_TICKS")
    env = rware_amd.WarehouseVecEnv(256, library=LIB, **kw)              # 16 workgroups: two rounds
    assert env.engines[0].info.stagger_ticks == 25 and env.engines[0].info.n_workgroups == 16
    orc = OracleVecEnv(256, **kw)
    assert np.array_equal(env.reset(seed=5)[0], orc.reset(seed=5))
    rng = np.random.default_rng(6)
    for t in range(25):
        a = rng.integers(0, 5, size=(256, 4), dtype=np.int32)
        obs, rew, term, _, _ = env.step(a)
        o2, r2, d2 = orc.step_autoreset(a, "next_step")
        assert np.array_equal(obs, o2) and np.array_equal(rew, r2) and np.array_equal(term, d2.astype(bool)), t
    env.close()


def test_wave_priority_is_a_create_time_rule_and_does_not_change_results(monkeypatch):
    """rw_info.wave_priority (round 6): per-step launches raise their wavefronts' priority up to the agent-phase barrier — on by rw_create's
    rule except for 13 .. 16 agents at sensor_range 1 and for steps with 200 MB of observations and more; RWARE_PRIO moves it.  A
    scheduling hint (nothing on the host threads of this build): the rule, the launch flag's way through every launch form, same results."""
    def wp(env_id, B, **extra):
        env = rware_amd.WarehouseVecEnv(B, library=LIB, **dict(rware_amd.env_kwargs(env_id), **extra))
        v = env.engines[0].info.wave_priority   # bit 0: per-step launches, bit 1: fused rollouts
        env.close()
        return v
    assert wp("rware-small-4ag-v1", 64) == 3 and wp("rware-small-12ag-v1", 8) == 3 and wp("rware-small-17ag-v1", 8) == 3
    # 13 .. 16 agents at one full round of 8-env workgroups (the emulated device has one CU: 8 workgroups, 64 envs): their rollouts gain,
    # their start-staggered per-step launches lose (below and above that batch they run on 4-env workgroups WITH the priority: next test)
    assert wp("rware-medium-13ag-v1", 64) == 2 and wp("rware-large-16ag-v1", 64) == 2
    assert wp("rware-large-16ag-v1", 8, sensor_range=2) == 3       # BASELINE config 5's shape
    monkeypatch.setenv("RWARE_PRIO", "0")
    assert wp("rware-small-4ag-v1", 64) == 2
    monkeypatch.setenv("RWARE_PRIO_ROLLOUT", "0")
    assert wp("rware-small-4ag-v1", 64) == 0
    monkeypatch.delenv("RWARE_PRIO_ROLLOUT")
    monkeypatch.setenv("RWARE_PRIO", "1")
    assert wp("rware-medium-13ag-v1", 64) == 3
    monkeypatch.delenv("RWARE_PRIO")
    # the caller's flag (RW_PRIO_OFF / RW_PRIO_ON; what make_pipelines passes for small sub-batches) wins over rule and hooks
    assert wp("rware-small-4ag-v1", 64, wave_priority=False) == 0 and wp("rware-medium-13ag-v1", 64, wave_priority=True) == 3
    monkeypatch.setenv("RWARE_PRIO", "1")
    assert wp("rware-small-4ag-v1", 64, wave_priority=False) == 0
    kw = rware_amd.env_kwargs("rware-small-4ag-v1")
    kw["max_steps"] = 9
    on = rware_amd.WarehouseVecEnv(32, library=LIB, **kw)
    monkeypatch.setenv("RWARE_PRIO", "0")
    off = rware_amd.WarehouseVecEnv(32, library=LIB, **kw)
    assert (on.engines[0].info.wave_priority & 1, off.engines[0].info.wave_priority & 1) == (1, 0)
    assert np.array_equal(on.reset(seed=2)[0], off.reset(seed=2)[0])
    acts = np.random.default_rng(8).integers(0, 5, size=(24, 32, 4), dtype=np.int32)
    for t in range(12):
        a, b = on.step(acts[t]), off.step(acts[t])
        assert all(np.array_equal(x, y) for x, y in zip(a[:4], b[:4])), t
    ra, rb = on.rollout(acts[12:]), off.rollout(acts[12:])       # (fused rollouts carry the flag by a rule of their own: both do here)
    assert all(np.array_equal(x, y) for x, y in zip(ra, rb))
    sa, sb = on.get_state(), off.get_state()
    assert all(np.array_equal(sa[k], sb[k]) for k in sa)
    on.close(); off.close()


def test_13_to_16_agents_step_on_4_env_workgroups_and_roll_out_on_8(monkeypatch):
    """Round 6: 13 .. 16 agents at sensor_range 1 — the per-step launches run on 4-env workgroups (one agent wavefront each) at raised
    wavefront priority below one full round of 8-env workgroups and between one and four rounds, on the start-staggered 8-env build at
    exactly one round and from four rounds on; the fused rollouts keep the 8-env build at every batch, so ONE engine launches its two
    kernels with different geometries.  The rule (the emulated device has one CU: a round is 8 workgroups), the hook, and per-step
    launches interleaved with fused rollouts against the oracle."""
    def geom(env_id, B):
        env = rware_amd.WarehouseVecEnv(B, library=LIB, **rware_amd.env_kwargs(env_id))
        i = env.engines[0].info
        out = (i.envs_per_workgroup, i.n_workgroups, i.wave_priority, i.stagger_ticks)
        env.close()
        return out
    assert geom("rware-large-16ag-v1", 32) == (4, 8, 3, 0)          # half a round of 8-env workgroups: 4-env, priority, no stagger
    assert geom("rware-large-16ag-v1", 64) == (8, 8, 2, 55)         # exactly one round: the staggered 8-env launch
    assert geom("rware-large-16ag-v1", 128) == (4, 32, 3, 0)        # two rounds
    assert geom("rware-large-16ag-v1", 256) == (8, 32, 2, 55)       # four rounds
    assert geom("rware-medium-13ag-v1", 32)[0] == 4 and geom("rware-small-14ag-v1", 128)[0] == 4
    assert geom("rware-tiny-14ag-v1", 32) == (8, 4, 3, 0)           # (the tiny warehouse has no 4-env build: 8-env, priority up to half a round)
    assert geom("rware-tiny-14ag-v1", 64) == (8, 8, 2, 55)
    # 9 .. 12 and 17 .. 19 agents: the 4-env build while its workgroups stay under one round (here: fewer than 8), the 8-env one from there on
    assert geom("rware-small-10ag-v1", 16) == (4, 4, 3, 0) and geom("rware-small-19ag-v1", 28)[0] == 4 and geom("rware-tiny-10ag-v1", 16)[0] == 8
    assert geom("rware-small-12ag-v1", 32)[0] == 8 and geom("rware-small-17ag-v1", 32)[0] == 8 and geom("rware-small-8ag-v1", 16)[0] == 8
    monkeypatch.setenv("RWARE_WIDE_E4", "0")
    assert geom("rware-large-16ag-v1", 32)[0] == 8
    monkeypatch.setenv("RWARE_WIDE_E4", "1")
    assert geom("rware-large-16ag-v1", 64) == (4, 16, 3, 0)
    monkeypatch.delenv("RWARE_WIDE_E4")
    # (9 .. 12 agents: the fused rollout follows onto the 4-env build; 13 .. 19: it stays on the 8-env one)
    for env_id, B in (("rware-large-16ag-v1", 32), ("rware-medium-13ag-v1", 128), ("rware-small-10ag-v1", 16), ("rware-small-19ag-v1", 16)):
        kw = rware_amd.env_kwargs(env_id)
        kw["max_steps"] = 9
        kw["reward_type"] = rware_amd.enums.enum_value(kw["reward_type"])
        N = kw["n_agents"]
        env = rware_amd.WarehouseVecEnv(B, library=LIB, **kw)
        assert env.engines[0].info.envs_per_workgroup == 4
        orc = OracleVecEnv(B, **kw)
        assert np.array_equal(env.reset(seed=4)[0], orc.reset(seed=4))
        rng = np.random.default_rng(0)
        for rnd in range(2):
            for t in range(5):
                a = rng.choice(5, size=(B, N), p=[.1, .55, .1, .1, .15]).astype(np.int32)
                o, r, d, _, _ = env.step(a)
                o2, r2, d2 = orc.step_autoreset(a, "next_step")
                assert np.array_equal(o, o2) and np.array_equal(r, r2) and np.array_equal(d, d2.astype(bool)), (env_id, rnd, t)
            acts = rng.choice(5, size=(6, B, N), p=[.1, .55, .1, .1, .15]).astype(np.int32)
            obs, rew, term = env.rollout(acts)        # the 8-env rollout build on the state the 4-env step kernel left, and back
            for t in range(6):
                o2, r2, d2 = orc.step_autoreset(acts[t], "next_step")
                assert np.array_equal(obs[t], o2) and np.array_equal(rew[t], r2) and np.array_equal(term[t], d2.astype(bool)), (env_id, rnd, t)
        st, so = env.get_state(), orc.get_state()
        assert all(np.array_equal(st[k], so[k]) for k in so), env_id
        env.close()


@pytest.mark.parametrize("env_id,extra,B,mode", [
    ("rware-small-4ag-v1", {"max_steps": 25}, 64, "next_step"),                     # 4 chunks on the emulation's 2 persistent workgroups
    ("rware-small-4ag-v1", {"max_steps": 25}, 48, "same_step"),                     # ragged: 2 + 1 chunks; terminal observations
    ("rware-small-4ag-v1", {"max_steps": 25}, 16, "disabled"),                      # one chunk: the pipeline never fills
    ("rware-small-4ag-v1", {"max_steps": 20, "reward_type": 2, "max_inactivity_steps": 11}, 80, "next_step"),
    ("rware-medium-6ag-hard-v1", {"max_steps": 20, "reward_type": 0}, 40, "next_step"),
    ("rware-large-16ag-v1", {"sensor_range": 2, "max_steps": 15}, 20, "same_step"),  # BASELINE config 5's shape: 4-env chunks
    ("rware-tiny-2ag-v1", {"max_steps": 25}, 96, "next_step"),                      # 32-env chunks
    ("rware-small-10ag-easy-v1", {"max_steps": 20}, 12, "same_step"),               # agent-count-static, run-time queue length, per-cell agent phases
    ("rware-large-16ag-v1", {"max_steps": 15}, 12, "next_step"),
])
def test_emulated_pipelined_build_matches_oracle(env_id, extra, B, mode):
    """The chunk-pipelined persistent build of the per-step kernel (rware_kernels.h "PIPE", opt-in: pipe=True): two workgroups walk
    the batch's chunks through two LDS buffers, the next chunk's agent phases beside this chunk's observation stores.  Resets
    (rw_reset, autoreset in both modes) go through the classic kernel on the same state."""
    kw = rware_amd.env_kwargs(env_id)
    kw.update(extra)
    kw["reward_type"] = rware_amd.enums.enum_value(kw["reward_type"])
    env = rware_amd.WarehouseVecEnv(B, autoreset_mode=mode, library=LIB, pipe=True, **kw)
    info = env.engines[0].info
    assert info.pipe_workgroups in (1, 2) and B % info.pipe_envs_per_workgroup == 0
    orc = OracleVecEnv(B, **kw)
    obs, _ = env.reset(seed=7)
    assert np.array_equal(obs, orc.reset(seed=7))
    rng = np.random.default_rng(3)
    for t in range(50):
        a = rng.choice(5, size=(B, kw["n_agents"]), p=[0.1, 0.55, 0.1, 0.1, 0.15]).astype(np.int32)
        obs, rew, term, trunc, inf = env.step(a)
        o2, r2, d2 = orc.step_autoreset(a, mode)
        assert np.array_equal(obs, o2), t
        assert np.array_equal(rew, r2) and np.array_equal(term, d2.astype(bool)), t
        if mode == "same_step" and d2.any():
            assert np.array_equal(inf["final_obs"][orc.final_mask], orc.final_obs[orc.final_mask]), t
    st, so = env.get_state(), orc.get_state()
    for k in so:
        assert np.array_equal(st[k], so[k]), k
    env.close()


def test_emulated_pipelined_build_is_opt_in():
    kw = rware_amd.env_kwargs("rware-small-4ag-v1")
    env = rware_amd.WarehouseVecEnv(64, library=LIB, **kw)
    assert env.engines[0].info.pipe_workgroups == 0      # the measured rule: never by default (profiles/EXPERIMENTS.md, round 5)
    env.close()
    env = rware_amd.WarehouseVecEnv(64, library=LIB, pipe=True, observation_type=2, **kw)
    assert env.engines[0].info.pipe_workgroups == 0      # no pipelined build for IMAGE observations: the classic kernel runs
    env.close()


SQUARE5 = dict(shelf_columns=3, column_height=3, shelf_rows=2, n_agents=5, msg_bits=0, sensor_range=2,
               request_queue_size=3, max_inactivity_steps=None, max_steps=12, reward_type=1)   # a 10 x 10 grid


@pytest.mark.parametrize("kw,extra,B,geom", [
    (dict(rware_amd.env_kwargs("rware-small-4ag-v1"), max_steps=15), dict(observation_type=2), 16, (0, 0)),          # exact IMAGE build
    (dict(rware_amd.env_kwargs("rware-medium-6ag-hard-v1"), max_steps=14, max_inactivity_steps=9), dict(observation_type=3), 8, (8, 128)),
    (dict(rware_amd.env_kwargs("rware-tiny-2ag-v1"), max_steps=20, sensor_range=2), dict(observation_type=3, image_observation_directional=False), 7, (4, 64)),
    # every layer incl. the two the reference writes with transposed indices (a square grid: no IndexError), rotated and north-up
    (SQUARE5, dict(observation_type=2, image_observation_layers=[3, 4, 0, 1, 2, 5, 6]), 8, (4, 64)),
    (SQUARE5, dict(observation_type=3, image_observation_layers=[4, 5, 3], image_observation_directional=False), 5, (4, 128)),
])
def test_emulated_image_terminal_observations_same_step(kw, extra, B, geom):
    """SAME_STEP autoreset with IMAGE / IMAGE_DICT observations: the image (and feature vectors) the terminating step itself produced
    (rware/warehouse.py:527-596, 722-744, 929-946) are kept as info["final_obs"] — round 5 closes the hole FLATTENED / DICT never had."""
    kw = dict(kw, reward_type=rware_amd.enums.enum_value(kw["reward_type"]))
    env = rware_amd.WarehouseVecEnv(B, autoreset_mode="same_step", library=LIB, envs_per_workgroup=geom[0], threads_per_workgroup=geom[1], **kw, **extra)
    orc = OracleVecEnv(B, **kw, **extra)
    n = gu.check_same_step_image_run(env, orc, B, kw["n_agents"], steps=45, seed=5)
    assert n > 0
    env.close()


def test_rw_multi_launcher_threads_overlap_the_enqueues(monkeypatch):
    """what rw_multi's thread mode is FOR, measured without eight GPUs — the emulation's launch is made a
    "null device" that holds the calling thread for 5 ms per enqueue (RWARE_EMU_LAUNCH_COST_US) and runs nothing.  One
    rw_multi_step_device call over 8 engines: with a launcher thread per engine the call returns in about ONE enqueue (<= 1.5 x
    what a single engine's call takes on this box right now, plus scheduling slack: the suite runs 6 workers on 8 cores), with
    the in-call loop in about eight."""
    import time
    from rware_amd import _capi
    kw = rware_amd.env_kwargs("rware-tiny-2ag-v1")
    res = {}

    def best_of(fn, n=10):
        best = 1e9
        for _ in range(n):
            t0 = time.perf_counter()
            fn()
            best = min(best, time.perf_counter() - t0)
        return best * 1e6

    for threads in ("1", "0"):
        monkeypatch.setenv("RWARE_MULTI_THREADS", threads)
        monkeypatch.delenv("RWARE_EMU_LAUNCH_COST_US", raising=False)
        env = rware_amd.WarehouseVecEnv(64, library=LIB, devices=[0] * 8, **kw)
        env.reset(seed=1)
        multi = _capi.MultiEngine(env.engines)
        bufs = [np.zeros((8, 2), np.int32) for _ in range(8)]
        ptrs = [x.ctypes.data for x in bufs]
        multi.step_device(ptrs)                                     # (threads up and spinning)
        monkeypatch.setenv("RWARE_EMU_LAUNCH_COST_US", "5000")
        res[threads] = best_of(lambda: multi.step_device(ptrs))
        res["one"] = best_of(lambda: env.engines[0].step_device(ptrs[0]))
        monkeypatch.delenv("RWARE_EMU_LAUNCH_COST_US")
        multi.close(); env.close()
    assert res["1"] <= 1.5 * res["one"] + 1500, res   # a launcher thread per engine: one enqueue's worth of wall time ...
    assert res["0"] >= 7.5 * 5000, res                # ... the in-call loop: eight
    assert res["1"] <= 0.4 * res["0"], res


def test_library_hooks_need_their_switch(monkeypatch):
    """The environment hooks (csrc/rware_hooks.h) are test / A-B instruments, not API: without RWARE_HOOKS=1 a process that happens to
    inherit one of them runs the measured rules."""
    kw = rware_amd.env_kwargs("rware-small-4ag-v1")
    monkeypatch.setenv("RWARE_STAGGER_TICKS", "40")
    monkeypatch.setenv("RWARE_PIPE", "1")
    env = rware_amd.WarehouseVecEnv(64, library=LIB, **kw)
    assert env.engines[0].info.stagger_ticks == 40 and env.engines[0].info.pipe_workgroups > 0     # conftest.py switched the hooks on
    env.close()
    monkeypatch.delenv("RWARE_HOOKS")
    env = rware_amd.WarehouseVecEnv(64, library=LIB, **kw)
    assert env.engines[0].info.stagger_ticks == 0 and env.engines[0].info.pipe_workgroups == 0
    env.close()


def test_store_floor_measurement_leaves_the_observations_alone():
    """rw_debug_store_floor (bench.py's `roofline.store_only_*`): launches that only write one step's observation bytes; RW_BUF_OBS is
    refreshed afterwards, the state is not touched."""
    kw = rware_amd.env_kwargs("rware-small-4ag-v1")
    env = rware_amd.WarehouseVecEnv(32, library=LIB, **kw)
    o0, _ = env.reset(seed=3)
    s0 = env.get_state()
    assert env.engines[0].debug_store_floor(2) > 0
    assert np.array_equal(env.observations(), o0)
    s1 = env.get_state()
    for k in s0:
        assert np.array_equal(s0[k], s1[k]), k
    env.close()
