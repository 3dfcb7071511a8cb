"""Host-side mirror of the reference interface: layout, observation length, registry ids/kwargs.
Cross-checked against the oracle's independent restatement and, where /root/reference exists,
against the reference itself."""
import os

import numpy as np
import pytest

import ref_runner as rr
import rware_oracle as orc

import rware_amd
from rware_amd.layout import layout_from_params, layout_from_str, obs_length


@pytest.mark.parametrize("cols,rows,height", [(1, 1, 8), (3, 1, 8), (3, 2, 8), (5, 2, 8), (5, 3, 8), (3, 3, 3), (7, 4, 2), (9, 1, 1)])
def test_layout_from_params_matches_oracle_restatement(cols, rows, height):
    lay = layout_from_params(cols, rows, height)
    hw, goals = orc.layout_from_params(cols, rows, height)
    assert lay.grid_size == ((height + 1) * rows + 2, 3 * cols + 1)        # warehouse.py:297-300
    assert np.array_equal(lay.highways, hw) and list(lay.goals) == goals
    assert lay.highways[-1].all() and lay.highways[:, 0].all()             # delivery row, left highway
    assert all(lay.highways[y, x] for x, y in lay.goals)


def test_layout_from_str():
    s = """
        .x.x.
        .x.x.
        ..g..
    """
    lay = layout_from_str(s)
    hw, goals = orc.layout_from_str(s)
    assert lay.grid_size == (3, 5) and np.array_equal(lay.highways, hw) and list(lay.goals) == goals == [(2, 2)]
    assert lay.n_shelves == 4
    with pytest.raises(AssertionError):
        layout_from_str("..\n...")
    with pytest.raises(AssertionError):
        layout_from_str("....")


def test_obs_length_formula():
    assert obs_length(1) == 71 and obs_length(2) == 183 and obs_length(3) == 351   # 8 + 7(2r+1)^2
    assert obs_length(1, msg_bits=2) == 71 + 2 * 9


def test_registry_ids_and_kwargs():
    ids = rware_amd.all_ids()
    assert len(ids) == 2 * 228 and len(set(ids)) == len(ids)
    kw = rware_amd.env_kwargs("rware-small-4ag-v1")
    assert kw == rware_amd.env_kwargs("rware-small-4ag-v2")
    assert (kw["shelf_rows"], kw["shelf_columns"], kw["n_agents"], kw["request_queue_size"]) == (2, 3, 4, 4)
    assert rware_amd.env_kwargs("rware-medium-6ag-hard-v1")["request_queue_size"] == 3
    assert rware_amd.env_kwargs("rware-tiny-1ag-hard-v2")["request_queue_size"] == 0
    assert rware_amd.env_kwargs("rware-large-19ag-easy-v2")["request_queue_size"] == 38
    for bad in ("rware-small-0ag-v1", "rware-small-20ag-v1", "rware-huge-2ag-v1", "rware-small-4ag-v3"):
        with pytest.raises(KeyError):
            rware_amd.env_kwargs(bad)


@pytest.mark.skipif(not rr.reference_available(), reason="/root/reference not present")
def test_registry_kwargs_equal_the_references_own():
    for env_id in rware_amd.all_ids(("v2",)):
        ref = rr.registry_kwargs(env_id)
        ours = rware_amd.env_kwargs(env_id)
        ref["reward_type"] = ref["reward_type"].value
        ours["reward_type"] = ours["reward_type"].value
        assert ref == ours, env_id


@pytest.mark.skipif(not rr.reference_available(), reason="/root/reference not present")
@pytest.mark.parametrize("env_id", ["rware-tiny-2ag-v2", "rware-small-4ag-v2", "rware-medium-6ag-hard-v2", "rware-large-16ag-v2"])
def test_layout_equals_the_references_own(env_id):
    env = rr.make_reference_env(env_id)
    kw = rware_amd.env_kwargs(env_id)
    lay = layout_from_params(kw["shelf_columns"], kw["shelf_rows"], kw["column_height"])
    assert lay.grid_size == tuple(env.grid_size)
    assert np.array_equal(lay.highways, env.highways) and list(lay.goals) == list(env.goals)


def test_enums_are_value_compatible():
    assert [a.value for a in rware_amd.Action] == [0, 1, 2, 3, 4]
    assert [d.name for d in rware_amd.Direction] == ["UP", "DOWN", "LEFT", "RIGHT"]
    assert rware_amd.RewardType.TWO_STAGE.value == 2 and rware_amd.ObservationType.FLATTENED.value == 1


def test_bench_rank_placement_partitions_physical_cores_per_numa_node(monkeypatch):
    """bench.pin_rank on a faked 2-socket box (2 NUMA nodes x 16 physical cores x 2 SMT threads, 8 GPUs, 4 per node): every
    rank gets whole physical cores of its GPU's node, disjoint from the other ranks' — the code path the 8-GPU scaling run
    takes, which no test box can exercise for real."""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench

    n_phys, nodes = 32, 2
    sib = {c: (c % n_phys, c % n_phys + n_phys) for c in range(2 * n_phys)}            # cpu c and c + 32 share a core
    node_of_core = lambda c: (c % n_phys) // (n_phys // nodes)                          # noqa: E731

    def fake_read(path):
        if path.startswith("/sys/bus/pci/devices/"):
            bus = int(path.split(":")[1], 16)
            return str(0 if bus < 4 else 1)                                              # GPUs 0-3 on node 0, 4-7 on node 1
        if path.startswith("/sys/devices/system/node/node"):
            k = int(path.split("node")[-1].split("/")[0])
            cs = [c for c in range(2 * n_phys) if node_of_core(c) == k]
            return ",".join(str(c) for c in cs)
        if "thread_siblings_list" in path:
            c = int(path.split("/cpu/cpu")[1].split("/")[0])
            return f"{sib[c][0]},{sib[c][1]}"
        return None

    class Props:
        def __init__(self, i):
            self.pci_domain_id, self.pci_bus_id, self.pci_device_id = 0, i, 0

    class FakeTorch:
        class cuda:
            @staticmethod
            def get_device_properties(i):
                return Props(i)

    masks = {}
    monkeypatch.setattr(bench, "_read", fake_read)
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(2 * n_phys)))
    monkeypatch.delenv("RWARE_BENCH_NO_PIN", raising=False)
    for r in range(8):
        monkeypatch.setattr(os, "sched_setaffinity", lambda pid, m, r=r: masks.__setitem__(r, set(m)))
        p = bench.pin_rank(FakeTorch, r, 8, lambda k: k)
        assert p["pinned"] and p["numa_node"] == (0 if r < 4 else 1) and p["physical_cores"] == 4, p
    for r in range(8):
        assert all(node_of_core(c) == (0 if r < 4 else 1) for c in masks[r])             # on the GPU's own node
        assert all((c + n_phys) % (2 * n_phys) in masks[r] for c in masks[r])            # whole cores: both SMT siblings
        for q in range(r):
            assert not (masks[r] & masks[q])                                              # disjoint between ranks
    # fewer physical cores than ranks on a node: report, do not pin
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: {0, 32})
    p = bench.pin_rank(FakeTorch, 1, 8, lambda k: k)
    assert p["pinned"] is False and "physical cores" in p["why"]


def test_make_pipelines_checks_its_arguments_before_touching_a_device():
    """rware_amd.make_pipelines(num_envs, n): the batch has to split evenly (the GPU side is covered by the -m gpu tests)."""
    with pytest.raises(ValueError):
        rware_amd.make_pipelines(10, 3, env_id="rware-tiny-2ag-v1")
    with pytest.raises(ValueError):
        rware_amd.make_pipelines(16, 0, env_id="rware-tiny-2ag-v1")
    p = rware_amd.Pipeline(env="e", stream="s", lo=8, hi=16)
    assert (p.env, p.stream, p.lo, p.hi) == ("e", "s", 8, 16)


def test_kernel_sources_hash_ignores_comments_and_layout():
    """bench.kernel_sources_sha() identifies the CODE a PMC traffic figure was measured on: comments and white space do not count
    (round 5: a comment-level edit behind the evidence pass orphaned profiles/pmc_traffic.json and `roofline.traffic` went null)."""
    import bench

    a = 'int f(int x) {  // add one\n    return x + 1; /* here */\n}\n\nconst char *s = "// kept /* kept */";\n'
    b = 'int f(int x) {\n  return x + 1;\n}\nconst char *s = "// kept /* kept */"; // gone\n'
    assert bench.strip_cxx_comments(a) == bench.strip_cxx_comments(b)
    assert bench.strip_cxx_comments(a) != bench.strip_cxx_comments(a.replace("x + 1", "x + 2"))
    assert '"// kept /* kept */"' in bench.strip_cxx_comments(a)
    assert len(bench.kernel_sources_sha()) == 16


def test_pmc_traffic_record_belongs_to_these_kernel_sources():
    """profiles/pmc_traffic.json — the rocprofv3 FETCH_SIZE / WRITE_SIZE passes behind `roofline.traffic` — must have been measured on
    the kernel sources in this tree, or bench.py reports `traffic: null` (`basis: "engine-bytes"`).  A kernel edit after the evidence
    pass turns THIS test red instead of silently dropping the field: re-run profiles/tools/final_pass.sh and commit the new record.
    (RWARE_ALLOW_STALE_PMC=1: development runs between two evidence passes.)"""
    import json

    import bench

    rec = json.load(open(os.path.join(bench.ROOT, "profiles", "pmc_traffic.json")))
    assert f"{bench.ENV_ID}:{bench.BATCH_PER_GPU}" in rec["entries"] and f"{bench.ENV_ID}:{bench.HBM_REGIME_BATCH}" in rec["entries"]
    if os.environ.get("RWARE_ALLOW_STALE_PMC") == "1":
        pytest.skip("RWARE_ALLOW_STALE_PMC=1")
    assert rec["kernel_sources_sha"] == bench.kernel_sources_sha(), (
        "profiles/pmc_traffic.json was measured on other kernel sources: run profiles/tools/final_pass.sh on the GPU box and commit its record")
    traffic, note = bench.pmc_traffic(bench.ENV_ID, bench.BATCH_PER_GPU, bench.kernel_sources_sha())
    assert note is None and 0.9 < traffic / (1433 * bench.BATCH_PER_GPU) < 1.15   # (engine bytes per env-step of small-4ag: DESIGN.md §4)
