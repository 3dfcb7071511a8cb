"""Generates tests/golden/events/events.npz: per step and env, the two event counts the engine can keep (RW_BUF_STAT_DELIVERIES /
RW_BUF_STAT_FAILED_MOVES), read off the UNMODIFIED reference's own objects while it replays the action streams of the golden traces
next to this directory (same kwargs, seeds, actions, NEXT_STEP autoreset, pinned tie-break — see generate_golden.py).
Run in the build container:  python tests/golden/generate_events.py

    deliveries   replacement draws of the step, one per delivered shelf (rware/warehouse.py:907-917)
    failed       agents that asked for FORWARD and whose req_action the step turned into NOOP (:843-846, :871-876)
(oracle/ref_runner.py ref_step_events).  The replay is checked against the stored trace (rewards, done) step by step, so the counts
belong to exactly those traces.  Stored as int8 [T][E] per trace: `<name>/deliveries`, `<name>/failed`.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_util as gu  # noqa: E402
import ref_runner as rr  # noqa: E402


def events_of(name):
    wh = rr.load_reference()
    meta, z = gu.load_fixture(name)
    kw = dict(meta["kwargs"])
    kw["reward_type"] = wh.RewardType(kw["reward_type"])
    kw["observation_type"] = wh.ObservationType(int(kw.get("observation_type", 1)))
    if "image_observation_layers" in kw:
        kw["image_observation_layers"] = [wh.ImageLayer(int(l)) for l in kw["image_observation_layers"]]
    E, T, M = meta["E"], meta["T"], int(kw.get("msg_bits", 0))
    envs = [wh.Warehouse(**kw) for _ in range(E)]
    for e, env in enumerate(envs):
        env.reset(seed=meta["seed"] + e)
    deliveries, failed = np.zeros((T, E), np.int8), np.zeros((T, E), np.int8)
    for t in range(T):
        for e, env in enumerate(envs):
            if z["was_reset"][t][e]:
                env.reset()
                continue
            a = z["actions"][t][e]
            a = [[int(v) for v in row] for row in a] if M else [int(v) for v in a]
            (_, r, d, _, _), nd, nf = rr.ref_step_events(env, a)
            assert np.array_equal(np.asarray(r, np.float32), z["rewards"][t][e]) and int(bool(d)) == int(z["done"][t][e]), (name, t, e)
            deliveries[t, e], failed[t, e] = nd, nf
    return deliveries, failed


if __name__ == "__main__":
    out, summary = {}, {}
    for name in gu.fixture_names():
        d, f = events_of(name)
        out[name + "/deliveries"], out[name + "/failed"] = d, f
        summary[name] = {"deliveries": int(d.sum()), "failed": int(f.sum())}
        print(f"{name}: deliveries {int(d.sum())} failed moves {int(f.sum())}")
    os.makedirs(os.path.join(HERE, "events"), exist_ok=True)
    np.savez_compressed(os.path.join(HERE, "events", "events.npz"), meta=json.dumps({
        "source": "semitable/robotic-warehouse @ /root/reference (rware 2.0.0), unmodified, pinned tie-break",
        "gymnasium": "standin" if rr.using_standin_gymnasium() else "real", "totals": summary}), **out)
    print("total deliveries", sum(v["deliveries"] for v in summary.values()), "failed moves", sum(v["failed"] for v in summary.values()))
