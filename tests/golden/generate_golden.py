"""Generates tests/golden/*.npz by running the UNMODIFIED reference (/root/reference/rware)
in the build container.  Run:  python tests/golden/generate_golden.py

Each fixture is a trace of E independent reference `Warehouse` instances (env e seeded
`seed + e`, the Gymnasium vector-env convention) driven by a stored action stream under
NEXT_STEP autoreset semantics: when env e returns done at step t, step t+1 is
`env.reset()` (no reseed — the PCG64 stream continues), its action is ignored, rewards are
0 and done is False.  The only deviation from the stock reference is the pinned
tie-break of oracle/ref_runner.py (lowest agent id among equal-depth predecessors).

Per step and env the fixture stores the full state (grid, agent SoA, queue, counters,
PCG64 state incl. the buffered 32-bit half), the FLATTENED observation, rewards and done.
gymnasium used for generation: the stand-in under oracle/gymnasium_standin (recorded in
the `meta` field) — real gymnasium is not installed in the build container.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ref_runner as rr  # noqa: E402

LAYOUT_STR = """
.......
...x...
..x.x..
.x...x.
..x.x..
...x...
.g...g.
"""  # /root/reference/README.md custom-layout example shape (x shelves, . corridors, g goals)

# name, env_id, extra ctor kwargs, E envs, T steps, seed
CASES = [
    ("tiny-2ag", "rware-tiny-2ag-v2", {}, 4, 650, 0),
    ("small-4ag", "rware-small-4ag-v2", {}, 4, 650, 1000),
    ("medium-6ag-hard", "rware-medium-6ag-hard-v2", {}, 3, 650, 2000),
    ("large-16ag-sr2", "rware-large-16ag-v2", {"sensor_range": 2}, 2, 600, 3000),
    ("tiny-4ag-easy-twostage", "rware-tiny-4ag-easy-v2", {"reward_type": 2}, 3, 400, 4000),
    ("small-8ag-global-inact", "rware-small-8ag-v2",
     {"reward_type": 0, "max_inactivity_steps": 60, "max_steps": 300}, 3, 700, 5000),
    ("small-3ag-normcoord-sr3", "rware-small-3ag-v2",
     {"normalised_coordinates": True, "sensor_range": 3}, 2, 300, 6000),
    ("layoutstr-3ag", None,
     {"shelf_columns": 0, "column_height": 0, "shelf_rows": 0, "n_agents": 3, "msg_bits": 0,
      "sensor_range": 1, "request_queue_size": 2, "max_inactivity_steps": None, "max_steps": 200,
      "reward_type": 1, "layout": LAYOUT_STR}, 3, 450, 7000),
    ("small-19ag", "rware-small-19ag-v2", {}, 2, 300, 8000),
    ("tiny-1ag-hard-q0", "rware-tiny-1ag-hard-v2", {}, 2, 120, 9000),
    # IMAGE / IMAGE_DICT observations (warehouse.py:527-596), default layers
    ("img-small-4ag-directional", "rware-small-4ag-v2", {"observation_type": 2}, 3, 260, 10000),
    ("img-tiny-3ag-northup-sr2", "rware-tiny-3ag-v2",
     {"observation_type": 2, "image_observation_directional": False, "sensor_range": 2}, 2, 200, 11000),
    ("imgdict-medium-6ag-hard", "rware-medium-6ag-hard-v2", {"observation_type": 3, "max_steps": 90}, 2, 200, 12000),
    # communication bits (warehouse.py:255-259, 660-667, 810-812): actions are [Action, bit, bit, ...] per agent
    ("msg2-small-4ag", "rware-small-4ag-v2", {"msg_bits": 2, "max_steps": 150}, 3, 330, 13000),
    ("msg3-tiny-3ag-sr2", "rware-tiny-3ag-v2", {"msg_bits": 3, "sensor_range": 2}, 2, 200, 14000),
    # AGENT_DIRECTION / AGENT_LOAD layers, written by the reference as layer[ag.x, ag.y] (warehouse.py:552,558): a square
    # 10 x 10 grid keeps the transposed index in bounds (every registered layout raises IndexError within a few steps)
    ("img-square-5ag-transposed-layers", None,
     {"shelf_columns": 3, "column_height": 3, "shelf_rows": 2, "n_agents": 5, "msg_bits": 0, "sensor_range": 2,
      "request_queue_size": 3, "max_inactivity_steps": None, "max_steps": 120, "reward_type": 1,
      "observation_type": 2, "image_observation_layers": [3, 4, 0, 2]}, 3, 300, 15000),
    # image observations together with communication bits: actions [Action, bit, bit], messages are stored but not shown (:527-596)
    ("img-msg2-tiny-3ag-8layers", "rware-tiny-3ag-v2",
     {"observation_type": 2, "msg_bits": 2, "sensor_range": 2, "max_steps": 80,
      "image_observation_layers": [0, 1, 2, 5, 6, 0, 2, 6]}, 2, 180, 17000),
    # widest window (11 x 11, 77 bits per row), many agents, a non-default column height, TWO_STAGE rewards
    ("sr5-12ag-colheight5-twostage", None,
     {"shelf_columns": 5, "column_height": 5, "shelf_rows": 2, "n_agents": 12, "msg_bits": 0, "sensor_range": 5,
      "request_queue_size": 6, "max_inactivity_steps": None, "max_steps": 150, "reward_type": 2}, 2, 320, 18000),
    # every image layer type at once (square grid: the transposed layers stay in bounds), communication bit on top
    ("img-square-all7-msg1", None,
     {"shelf_columns": 3, "column_height": 3, "shelf_rows": 2, "n_agents": 4, "msg_bits": 1, "sensor_range": 1,
      "request_queue_size": 2, "max_inactivity_steps": 40, "max_steps": 100, "reward_type": 0,
      "observation_type": 2, "image_observation_layers": [0, 1, 2, 3, 4, 5, 6]}, 3, 240, 19000),
    # round 3: the builds added for odd agent counts, the large warehouse and the 2-agent tasks, pinned to the reference as well
    ("small-7ag-hard", "rware-small-7ag-hard-v2", {"max_steps": 150}, 2, 320, 20000),
    ("large-4ag", "rware-large-4ag-v2", {"max_steps": 200}, 2, 320, 21000),
    ("medium-2ag-easy", "rware-medium-2ag-easy-v2", {"max_steps": 150}, 4, 320, 22000),
    ("imgdict-square-5ag-transposed-northup", None,
     {"shelf_columns": 3, "column_height": 3, "shelf_rows": 2, "n_agents": 5, "msg_bits": 0, "sensor_range": 1,
      "request_queue_size": 3, "max_inactivity_steps": None, "max_steps": 90, "reward_type": 2,
      "observation_type": 3, "image_observation_directional": False, "image_observation_layers": [4, 5, 3]}, 2, 200, 16000),
]


def gen_case(name, env_id, extra, E, T, seed):
    wh = rr.load_reference()
    kw = rr.registry_kwargs(env_id) if env_id else {}
    kw.update(extra)
    kw_json = dict(kw)
    kw_json["reward_type"] = int(getattr(kw["reward_type"], "value", kw["reward_type"]))
    kw["reward_type"] = wh.RewardType(kw_json["reward_type"])
    obs_type = int(kw_json.get("observation_type", 1))
    kw["observation_type"] = wh.ObservationType(obs_type)
    if "image_observation_layers" in kw:
        kw["image_observation_layers"] = [wh.ImageLayer(int(l)) for l in kw_json["image_observation_layers"]]

    def obs_arrays(o):
        """(obs, features) of one env in array form for the configured observation type."""
        if obs_type == 3:
            return (np.stack([a["image"] for a in o]).astype(np.float32),
                    np.stack([a["features"] for a in o]).astype(np.float32))
        return rr.obs_array(o), None
    envs = [wh.Warehouse(**kw) for _ in range(E)]
    N = envs[0].n_agents
    M = int(kw_json.get("msg_bits", 0))
    state_keys = ("grid", "agent_x", "agent_y", "agent_dir", "agent_carry", "agent_delivered",
                  "queue", "steps", "inactive", "rng") + (("agent_msg",) if M else ())
    pol = np.random.default_rng(seed + 77)
    rec = {k: [] for k in state_keys + ("obs", "features", "rewards", "done", "actions", "was_reset")}

    def record(snaps, obs, rew, done, acts, was_reset):
        for k in state_keys:
            rec[k].append(np.stack([s[k] for s in snaps]))
        rec["obs"].append(np.stack([o[0] for o in obs]))
        if obs_type == 3:
            rec["features"].append(np.stack([o[1] for o in obs]))
        rec["rewards"].append(np.asarray(rew, np.float32))
        rec["done"].append(np.asarray(done, np.uint8))
        rec["actions"].append(np.asarray(acts, np.int8))
        rec["was_reset"].append(np.asarray(was_reset, np.uint8))

    obs0_pairs = [obs_arrays(env.reset(seed=seed + e)[0]) for e, env in enumerate(envs)]
    obs0 = [o[0] for o in obs0_pairs]
    init = {k: np.stack([rr.snapshot(env)[k] for env in envs]) for k in state_keys}
    prev_done = [False] * E
    deliveries = 0
    for t in range(T):
        phase = (t // 100) % 3
        acts, obs, rew, done, snaps, was_reset = [], [], [], [], [], []
        for e, env in enumerate(envs):
            if phase == 0:
                a = rr.scripted_actions(env, pol)
            elif phase == 1:
                a = [int(v) for v in pol.integers(0, 5, size=N)]
            else:
                a = [int(v) for v in pol.choice(5, size=N, p=[0.1, 0.6, 0.1, 0.1, 0.1])]
            if M:  # [Action, message bits...] per agent
                a = [[int(v)] + [int(b) for b in pol.integers(0, 2, size=M)] for v in a]
            acts.append(a)
            if prev_done[e]:
                o, _ = env.reset()
                r, d = [0.0] * N, False
                was_reset.append(1)
            else:
                o, r, d, _, _ = rr.ref_step(env, a)
                was_reset.append(0)
            deliveries += sum(r)
            obs.append(obs_arrays(o))
            rew.append(r)
            done.append(d)
            snaps.append(rr.snapshot(env))
            prev_done[e] = bool(d)
        record(snaps, obs, rew, done, acts, was_reset)

    out = {k: np.stack(v) for k, v in rec.items() if v}
    ids_max = max(int(out["grid"].max()), 1)
    out["grid"] = out["grid"].astype(np.uint8 if ids_max < 256 else np.int16)
    for k in ("agent_x", "agent_y", "agent_dir", "agent_delivered") + (("agent_msg",) if M else ()):
        out[k] = out[k].astype(np.int8)
    for k in ("agent_carry", "queue"):
        out[k] = out[k].astype(np.int16)
    if not kw_json.get("normalised_coordinates"):
        assert np.array_equal(out["obs"].astype(np.float16).astype(np.float32), out["obs"])
        out["obs"] = out["obs"].astype(np.float16)
        obs0s = np.stack(obs0).astype(np.float16)
    else:
        obs0s = np.stack(obs0)
    meta = {
        "name": name, "env_id": env_id, "kwargs": kw_json, "E": E, "T": T, "seed": seed,
        "autoreset": "next_step", "tie_break": "lowest_agent_id",
        "gymnasium": "standin" if rr.using_standin_gymnasium() else "real",
        "reference": "semitable/robotic-warehouse @ /root/reference (rware 2.0.0)",
        "deliveries": float(deliveries),
    }
    path = os.path.join(HERE, f"{name}.npz")
    extra_arrays = {"features0": np.stack([o[1] for o in obs0_pairs])} if obs_type == 3 else {}
    np.savez_compressed(path, meta=json.dumps(meta), obs0=obs0s, **extra_arrays,
                        **{f"init_{k}": v for k, v in init.items()}, **out)
    print(f"{name}: E={E} T={T} deliveries={deliveries} resets={int(out['was_reset'].sum())} "
          f"-> {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    only = set(sys.argv[1:])
    for case in CASES:
        if not only or case[0] in only:
            gen_case(*case)
    print("tie stats: calls", rr.TieStats.calls, "ties", rr.TieStats.ties,
          "unpatched-networkx disagreements", rr.TieStats.disagree)
