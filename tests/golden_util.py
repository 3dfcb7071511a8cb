"""Shared replay harness: drives a backend through a golden fixture and checks every field."""
import glob
import json
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
STATE_FIELDS = ("grid", "agent_x", "agent_y", "agent_dir", "agent_carry", "agent_delivered",
                "queue", "steps", "inactive", "rng")


def fixture_names():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


def load_fixture(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
    meta = json.loads(str(z["meta"]))
    return meta, z


def ctor_kwargs(meta):
    return dict(meta["kwargs"])


def assert_state_equal(got: dict, z, t, prefix=""):
    for k in STATE_FIELDS + (("agent_msg",) if "agent_msg" in z else ()):
        want = z[prefix + k] if t is None else z[k][t]
        g = np.asarray(got[k])
        assert g.shape == want.shape, (k, t, g.shape, want.shape)
        if not np.array_equal(g.astype(np.int64) if k != "rng" else g, want.astype(np.int64) if k != "rng" else want):
            bad = np.argwhere(g.astype(np.int64) != want.astype(np.int64))[:5]
            raise AssertionError(f"state field {k} differs at step {t}: first idx {bad.tolist()}")


def replay(backend, meta, z, steps=None):
    """backend: .reset(seed) -> obs ; .step_autoreset(actions, mode) -> (obs, rew, done) ; .get_state()"""
    def split(o):  # IMAGE_DICT backends return (image, features)
        return (o[0], o[1]) if isinstance(o, tuple) else (o, None)

    obs, feat = split(backend.reset(seed=meta["seed"]))
    assert np.array_equal(np.asarray(obs, np.float32), z["obs0"].astype(np.float32)), "reset obs"
    if "features0" in z:
        assert np.array_equal(np.asarray(feat, np.float32), z["features0"]), "reset features"
    assert_state_equal(backend.get_state(), z, None, prefix="init_")
    T = meta["T"] if steps is None else min(steps, meta["T"])
    for t in range(T):
        obs, rew, done = backend.step_autoreset(z["actions"][t].astype(np.int32), "next_step")
        obs, feat = split(obs)
        if "features" in z:
            assert np.array_equal(np.asarray(feat, np.float32), z["features"][t]), f"features t={t}"
        assert np.array_equal(np.asarray(rew, np.float32), z["rewards"][t]), f"rewards t={t}"
        assert np.array_equal(np.asarray(done).astype(np.uint8), z["done"][t]), f"done t={t}"
        assert_state_equal(backend.get_state(), z, t)
        o = np.asarray(obs, np.float32)
        w = z["obs"][t].astype(np.float32)
        if not np.array_equal(o, w):
            bad = np.argwhere(o != w)[:5]
            raise AssertionError(f"obs differs at step {t}: {bad.tolist()}")
    return T


def check_same_step_image_run(env, orc, B, N, steps, seed):
    """Steps `env` (IMAGE / IMAGE_DICT, same_step autoreset) beside the oracle: observations, rewards, flags every step, and the
    terminal observation of every env that ended an episode (info["final_obs"], rows info["_final_obs"]).  Returns how many
    terminal observations were compared."""
    def same(a, b):
        if isinstance(a, dict):
            return np.array_equal(a["image"], b[0]) and np.array_equal(a["features"], b[1])
        return np.array_equal(a, b)

    assert same(env.reset(seed=seed)[0], orc.reset(seed=seed))
    rng = np.random.default_rng(3)
    n_final = 0
    for t in range(steps):
        a = rng.choice(5, size=(B, N), p=[0.1, 0.55, 0.1, 0.1, 0.15]).astype(np.int32)
        obs, rew, term, trunc, info = env.step(a)
        o2, r2, d2 = orc.step_autoreset(a, "same_step")
        assert same(obs, o2), t
        assert np.array_equal(rew, r2) and np.array_equal(term, d2.astype(bool)), t
        assert ("final_obs" in info) == bool(d2.any()), t
        if d2.any():
            m = orc.final_mask
            assert np.array_equal(info["_final_obs"], m), t
            f, fo = info["final_obs"], orc.final_obs
            if isinstance(f, dict):
                assert np.array_equal(f["image"][m], fo[0][m]), (t, "image")
                assert np.array_equal(f["features"][m], fo[1][m]), (t, "features")
            else:
                assert np.array_equal(f[m], fo[m]), t
            n_final += int(m.sum())
    return n_final
