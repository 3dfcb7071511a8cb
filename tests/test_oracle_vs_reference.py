"""Live re-check of the oracle against the unmodified reference — runs only where
/root/reference exists (the build container); skipped on the GPU box."""
import numpy as np
import pytest

import ref_runner as rr
from rware_oracle import OracleVecEnv

pytestmark = pytest.mark.skipif(not rr.reference_available(), reason="/root/reference not present")


@pytest.mark.parametrize("env_id,extra", [
    ("rware-tiny-2ag-v2", {}),
    ("rware-small-4ag-v2", {}),
    ("rware-medium-6ag-hard-v2", {}),
    ("rware-large-16ag-v2", {"sensor_range": 2}),
])
def test_live_rollout(env_id, extra):
    kw = rr.registry_kwargs(env_id)
    kw.update(extra)
    env = rr.make_reference_env(None, **kw)
    orc = OracleVecEnv(1, **kw)
    seed = 4242
    obs, _ = env.reset(seed=seed)
    o2 = orc.reset(seed=seed)
    assert np.array_equal(rr.obs_array(obs), o2[0])
    pol = np.random.default_rng(7)
    for t in range(250):
        a = rr.scripted_actions(env, pol) if t % 2 else list(pol.choice(5, size=env.n_agents, p=[.1, .6, .1, .1, .1]))
        (obs, r, d, _, _), n_deliv, n_failed = rr.ref_step_events(env, a)
        c0 = int(orc.stat_deliveries[0]), int(orc.stat_failed_moves[0])
        r2, d2 = orc.step(np.array(a)[None])
        # the event counters (RW_BUF_STAT_*): this step's deliveries and failed moves as the reference's own objects show them
        assert (int(orc.stat_deliveries[0]) - c0[0], int(orc.stat_failed_moves[0]) - c0[1]) == (n_deliv, n_failed), t
        snap, st = rr.snapshot(env), orc.get_state()
        for k, v in snap.items():
            assert np.array_equal(np.asarray(v).reshape(-1), st[k][0].reshape(-1)), (k, t)
        assert np.array_equal(rr.obs_array(obs), orc.obs()[0])
        assert np.array_equal(np.asarray(r, np.float32), r2[0]) and bool(d) == bool(d2[0])


def test_registry_has_228_ids_and_v2_suffix():
    rr.load_reference()
    import gymnasium, rware  # noqa
    ids = [k for k in gymnasium.registry if k.startswith("rware-")]
    assert len(ids) == 228 and all(i.endswith("-v2") for i in ids)


@pytest.mark.parametrize("env_id,layer", [
    ("rware-tiny-2ag-v2", 3), ("rware-small-4ag-v2", 3), ("rware-small-4ag-v2", 4), ("rware-medium-6ag-hard-v2", 4),
])
def test_transposed_image_layers_raise_indexerror_at_the_same_step(env_id, layer):
    """AGENT_DIRECTION (3) / AGENT_LOAD (4) are written as layer[ag.x, ag.y] on an (H, W) array (:552, :558): on the
    registered layouts (H > W) the reference raises IndexError as soon as an agent — a loaded one for AGENT_LOAD —
    stands at y >= W.  The oracle reports it for the same reset()/step() call and matches every obs before it."""
    wh = rr.load_reference()
    hit = 0
    for seed in range(6):
        kw = rr.registry_kwargs(env_id)
        kw.update(observation_type=wh.ObservationType.IMAGE,
                  image_observation_layers=[wh.ImageLayer.AGENTS, wh.ImageLayer(layer)])
        env = wh.Warehouse(**kw)
        okw = {k: getattr(v, "value", v) for k, v in kw.items()}
        okw["image_observation_layers"] = [2, layer]
        orc = OracleVecEnv(1, **okw)

        def attempt(f):
            try:
                return f(), False
            except IndexError:
                return None, True

        (obs, e1), (o2, e2) = attempt(lambda: env.reset(seed=seed)[0]), attempt(lambda: orc.reset(seed=seed))
        assert e1 == e2, ("reset", seed)
        rng = np.random.default_rng(seed)
        t = 0
        while not e1 and t < 300:
            assert np.array_equal(np.stack(obs)[None], o2), (seed, t)
            a = rng.integers(0, 5, size=env.n_agents)
            (res, e1), (r2, e2) = attempt(lambda: rr.ref_step(env, list(a))), attempt(lambda: (orc.step(a[None].astype(np.int32)), orc.obs()))
            assert e1 == e2, (seed, t)
            if not e1:
                obs, o2 = res[0], r2[1]
                if res[2]:
                    break
            t += 1
        hit += int(e1)
    assert hit >= 3   # the IndexError is the common case, not a corner


@pytest.mark.parametrize("env_id,extra", [
    ("rware-tiny-2ag-v2", {}),
    ("rware-small-4ag-v2", {"msg_bits": 2, "sensor_range": 2}),
    ("rware-tiny-3ag-v2", {"normalised_coordinates": True}),
])
def test_dict_observations_of_the_engine_match_the_reference(env_id, extra):
    """ObservationType.DICT (:676-720) end to end: the product engine (host-thread emulation build) against the
    live reference, every key and leaf of every agent's nested dict."""
    import engine_backend as eb
    import rware_amd

    wh = rr.load_reference()
    kw = rr.registry_kwargs(env_id)
    kw.update(extra)
    B, seed = 3, 77
    refs = [wh.Warehouse(**dict(kw, observation_type=wh.ObservationType.DICT)) for _ in range(B)]
    ekw = {k: getattr(v, "value", v) for k, v in kw.items()}
    env = rware_amd.WarehouseVecEnv(B, library=eb.build_emu(), observation_type=rware_amd.ObservationType.DICT,
                                    autoreset_mode="disabled", envs_per_workgroup=4, threads_per_workgroup=64, **ekw)
    M = kw["msg_bits"]

    def check(got, want):
        assert type(got) is type(want) and set(got) == set(want)
        for part in ("self",):
            for k, v in want[part].items():
                g = got[part][k]
                assert (np.array_equal(g, v) and g.dtype == v.dtype) if isinstance(v, np.ndarray) else g == v, k
        assert len(got["sensors"]) == len(want["sensors"])
        for cg, cw in zip(got["sensors"], want["sensors"]):
            assert list(cg) == list(cw)          # same keys in the same order
            for k, v in cw.items():
                assert cg[k] == (list(v) if isinstance(v, np.ndarray) else v), k

    obs, _ = env.reset(seed=seed)
    for b, r in enumerate(refs):
        want, _ = r.reset(seed=seed + b)
        for g, w in zip(env.unbatch_dict_obs(obs, b), want):
            check(g, w)
    rng = np.random.default_rng(0)
    for t in range(60):
        a = rng.integers(0, 5, size=(B, refs[0].n_agents))
        full = a if not M else np.concatenate([a[..., None], rng.integers(0, 2, size=(B, refs[0].n_agents, M))], axis=-1)
        obs, rew, term, trunc, _ = env.step(full)
        for b, r in enumerate(refs):
            want, rr_, d, _, _ = rr.ref_step(r, [list(x) for x in full[b]] if M else list(full[b]))
            for g, w in zip(env.unbatch_dict_obs(obs, b), want):
                check(g, w)
            assert np.array_equal(np.asarray(rr_, np.float32), rew[b])
    env.close()


def test_cpu_baseline_legs_report_cores_and_model():
    """oracle/cpu_baseline.py — what bench.py prints as `cpu_baseline`: both legs (the unmodified reference step and the
    C port), one process and several, on a bounded sample; every case states the cores used and the CPU model."""
    import cpu_baseline as cb

    assert cb.host_cores() >= 1 and isinstance(cb.cpu_model(), str) and cb.cpu_model()
    one = cb.time_port("rware-small-4ag-v1", 0.3, 1, b=64)
    two = cb.time_port("rware-small-4ag-v1", 0.3, 2, b=64)
    assert one > 1e4 and two > 1e4                      # agent-steps/s of the C port: millions per core; far above 1e4 anywhere
    ref1, standin = cb.time_reference("rware-small-4ag-v1", 0.5, 1)
    ref2, _ = cb.time_reference("rware-small-4ag-v1", 0.5, 2)
    assert 1e2 < ref1 < one and 1e2 < ref2              # the pure-Python step: thousands per core
    assert isinstance(standin, bool)


def test_staged_reference_is_the_unmodified_tree():
    """oracle/make_ref.sh stages rware/{__init__,warehouse}.py byte for byte into the git-ignored oracle/_ref/
    (what bench.py's cpu_baseline leg imports on the GPU box): same sha256 as the files under /root/reference."""
    import hashlib
    import os
    import subprocess

    tree = "/root/reference"
    if not os.path.isfile(os.path.join(tree, "rware", "warehouse.py")):
        pytest.skip("no reference tree to compare the staged copy with")
    here = os.path.dirname(os.path.abspath(rr.__file__))
    subprocess.check_call(["bash", os.path.join(here, "make_ref.sh")])
    lines = open(os.path.join(rr.STAGED_ROOT, "MANIFEST.sha256")).read().split("\n")
    seen = 0
    for line in filter(None, lines):
        sha, rel = line.split()
        for root in (rr.STAGED_ROOT, tree):
            with open(os.path.join(root, rel), "rb") as f:
                assert hashlib.sha256(f.read()).hexdigest() == sha, (root, rel)
        seen += 1
    assert seen == 2
    tracked = subprocess.run(["git", "ls-files", "oracle/_ref"], cwd=os.path.dirname(here), capture_output=True, text=True).stdout
    assert tracked.strip() == "", "the staged reference must stay out of git history"


def test_opt_in_registries_parse_to_the_reference_kwargs():
    """`image_registration()` (rware/__init__.py:42-80) and `full_registration()` (:83-175) add millions of ids; the engine's
    registry parses their grammar instead of listing them.  Every id of the first and a 1-in-3000 sample of the second (the
    reference's own loops, with `register` swapped for a recorder) must map to exactly the kwargs the reference registers."""
    import zlib

    import rware_amd
    from rware_amd.enums import enum_value

    rr.load_reference()
    import rware

    got = {}

    def recorder(id, entry_point=None, kwargs=None, **_):
        if recorder.everything or zlib.crc32(id.encode()) % 3000 == 0:
            got[id] = dict(kwargs)

    orig = rware.register
    rware.register = recorder
    try:
        recorder.everything = True
        rware.image_registration()
        n_image = len(got)
        recorder.everything = False
        rware.full_registration()
    finally:
        rware.register = orig
    assert n_image == 4 * 3 * 19 * 4 and len(got) > n_image + 1000   # (img, img-Nd, imgdict, imgdict-Nd) x 228; plus the sample
    for env_id, want in got.items():
        for ver in ("-v2", "-v1"):
            kw = rware_amd.env_kwargs(env_id[:-3] + ver)
            kw.setdefault("observation_type", 1)                     # FLATTENED / directional are the constructor defaults
            kw.setdefault("image_observation_directional", True)
            norm = lambda v: v if v is None or isinstance(v, bool) else enum_value(v)  # noqa: E731
            assert {k: norm(v) for k, v in kw.items()} == {k: norm(v) for k, v in want.items()}, env_id
    for bad in ("rware-Nd-tiny-2ag-v2", "rware-imgdict-2s-tiny-8h-2ag-v2", "rware-img-tiny-20ag-v2", "rware-2x4-8h-2ag-2req-indiv-v2",
                "rware-6s-tiny-8h-2ag-v2", "rware-tiny-16h-2ag-v2"):
        with pytest.raises(KeyError):
            rware_amd.env_kwargs(bad)


@pytest.mark.parametrize("env_id,extra", [
    ("rware-tiny-2ag-v2", {}),
    ("rware-small-4ag-v2", {}),
    (None, dict(shelf_columns=3, column_height=3, shelf_rows=2, n_agents=5, msg_bits=0, sensor_range=1, request_queue_size=3,
                max_inactivity_steps=None, max_steps=500, reward_type=1)),   # a square 10 x 10 grid: the transposed layers stay inside
])
def test_global_image_matches_the_reference(env_id, extra):
    """`get_global_image` (rware/warehouse.py:966-1040) rebuilt from the batched state: every layer type, the default pair,
    padding — against the live reference at the same state (the oracle steps beside it and supplies get_state())."""
    from rware_amd.vector_env import global_image_from_state

    kw = rr.registry_kwargs(env_id) if env_id else {}
    kw.update(extra)
    env = rr.make_reference_env(None, **kw)
    wh = rr.load_reference()
    orc = OracleVecEnv(1, **kw)
    env.reset(seed=99)
    orc.reset(seed=99)
    pol = np.random.default_rng(3)
    square = env.grid_size[0] == env.grid_size[1]
    sets = [[0, 5], [0, 1, 2, 5, 6]] + ([[0, 1, 2, 3, 4, 5, 6], [3], [4]] if square else [])
    for t in range(80):
        a = rr.scripted_actions(env, pol) if t % 3 else list(pol.choice(5, size=env.n_agents, p=[.1, .5, .1, .1, .2]))
        rr.ref_step(env, a)
        orc.step(np.array(a)[None])
        if t % 8:
            continue
        st = orc.get_state()
        for ls in sets:
            want = env.get_global_image(image_layers=[wh.ImageLayer(l) for l in ls], recompute=True)
            got = global_image_from_state(st, env.goals, ls)
            assert got.shape == (1,) + want.shape and np.array_equal(got[0], want), (t, ls)
        shape = (2, env.grid_size[0] + 3, env.grid_size[1] + 2)
        want = env.get_global_image(image_layers=[wh.ImageLayer(0), wh.ImageLayer(5)], recompute=True, pad_to_shape=shape)
        assert np.array_equal(global_image_from_state(st, env.goals, [0, 5], shape)[0], want)
    if not square:  # rectangular grids: the reference raises IndexError once an agent stands at x >= H or y >= W
        bad = {k: v.copy() for k, v in orc.get_state().items()}
        bad["agent_y"][0, 0] = env.grid_size[1]
        with pytest.raises(IndexError):
            global_image_from_state(bad, env.goals, [3])


@pytest.mark.parametrize("case", range(24))
def test_oracle_matches_the_live_reference_on_random_shapes(case):
    """24 warehouses the fixtures do not hold — random shelf rows / columns / column height, 1..10 agents, any queue length,
    sensor range 1..4, all three reward types, optional inactivity limit and normalised coordinates — each 140 steps of the
    unmodified reference (pinned tie-break) against the C oracle: state incl. the PCG64 stream, observations, rewards, done."""
    g = np.random.default_rng(5000 + case)
    rows, cols, height = int(g.integers(1, 4)), int(g.choice([1, 3, 5])), int(g.integers(1, 9))
    n_agents = int(g.integers(1, 11))
    shelves = rows * cols * 2 * height - 0  # (upper bound; the goal block removes some)
    kw = dict(shelf_columns=cols, column_height=height, shelf_rows=rows, n_agents=n_agents, msg_bits=0,
              sensor_range=int(g.integers(1, 5)), request_queue_size=int(g.integers(0, max(1, min(2 * n_agents, shelves // 3)) + 1)),
              max_inactivity_steps=(None if g.random() < 0.6 else int(g.integers(15, 60))), max_steps=int(g.integers(40, 120)),
              reward_type=int(g.integers(0, 3)), normalised_coordinates=bool(g.random() < 0.25))
    wh = rr.load_reference()
    env = rr.make_reference_env(None, **dict(kw, reward_type=wh.RewardType(kw["reward_type"])))   # (the reference compares enum members)
    if env.n_agents > (env.grid_size[0] * env.grid_size[1]) // 2:
        pytest.skip("more agents than this tiny grid can hold comfortably")
    orc = OracleVecEnv(1, **kw)
    seed = 900 + case
    try:
        obs, _ = env.reset(seed=seed)
    except ValueError:
        pytest.skip("a layout without shelves (one column: it is the goal column): the reference's own reset() raises")
    if kw["request_queue_size"] >= len(env.shelfs):
        pytest.skip("as many requests as shelves: the reference's own replacement draw has no candidates (ValueError)")
    assert np.array_equal(rr.obs_array(obs), orc.reset(seed=seed)[0])
    pol = np.random.default_rng(seed)
    done_prev = False
    for t in range(140):
        a = rr.scripted_actions(env, pol) if (t // 35) % 2 == 0 else [int(v) for v in pol.choice(5, size=env.n_agents, p=[.1, .55, .1, .1, .15])]
        if done_prev:  # the reference's caller resets; the oracle's next_step autoreset does the same on this step
            obs, _ = env.reset()
            o2, r2, d2 = orc.step_autoreset(np.array(a)[None], "next_step")
            r, d = [0.0] * env.n_agents, False
        else:
            (obs, r, d, _, _), n_deliv, n_failed = rr.ref_step_events(env, a)
            c0 = int(orc.stat_deliveries[0]), int(orc.stat_failed_moves[0])
            o2, r2, d2 = orc.step_autoreset(np.array(a)[None], "next_step")
            assert (int(orc.stat_deliveries[0]) - c0[0], int(orc.stat_failed_moves[0]) - c0[1]) == (n_deliv, n_failed), (t, kw)
        snap, st = rr.snapshot(env), orc.get_state()
        for k, v in snap.items():
            assert np.array_equal(np.asarray(v).reshape(-1), st[k][0].reshape(-1)), (k, t, kw)
        assert np.array_equal(rr.obs_array(obs), o2[0]), (t, kw)
        assert np.array_equal(np.asarray(r, np.float32), r2[0]) and bool(d) == bool(d2[0]), (t, kw)
        done_prev = bool(d)
