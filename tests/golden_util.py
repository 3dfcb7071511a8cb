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
