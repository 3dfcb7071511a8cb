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
        obs, r, d, _, _ = rr.ref_step(env, a)
        r2, d2 = orc.step(np.array(a)[None])
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
