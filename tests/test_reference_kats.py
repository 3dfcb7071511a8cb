"""The reference's known-answer tests for the step path, restated table-driven and replayed
against three implementations: the CPU oracle, the product sources under host-thread emulation,
and (GPU-marked) the real HIP engine.

Source of every expectation: /root/reference/tests/test_movement.py (:50-620, moves, wall clamps,
head-on swaps, chains, cycles, turn tables, carrying, pick-up/unload), tests/test_goals.py
(:99-184, delivery + reward types), tests/test_env.py (:42-68 grid size, :336-403 termination).
Fixture env of the reference: Warehouse(3, 8, 3, N, 0, 1, 5, None, None, GLOBAL) = 29 x 10 grid.
"""
import numpy as np
import pytest

from kat_env import (DOWN, FORWARD, GLOBAL, INDIVIDUAL, LEFT, NOOP, RIGHT, TOGGLE, TURN_LEFT,
                     TURN_RIGHT, TWO_STAGE, UP, KatEnv)

# "-static": the same scenarios on the exact-shape kernel build of rware-small-4ag (agent phases in registers, cross-lane
# exchange) — workgroups of 16 envs, and on the GPU also the 8-env build picked for small batches
BACKENDS = ["oracle", "emu", "emu-static", pytest.param("gpu", marks=pytest.mark.gpu),
            pytest.param("gpu-static", marks=pytest.mark.gpu), pytest.param("gpu-static8", marks=pytest.mark.gpu)]
GENERIC_ONLY = ["oracle", "emu", pytest.param("gpu", marks=pytest.mark.gpu)]


def make(backend, n_agents, reward=GLOBAL, cols=3, height=8, rows=3, queue=5, max_inact=None, max_steps=None, **kw):
    lib = None
    if backend.startswith("emu"):
        from engine_backend import build_emu
        lib = build_emu()
    static = None
    if backend.endswith("-static") or backend.endswith("-static8"):
        if n_agents > 4:
            pytest.skip("the exact-shape build under test has 4 agents")
        static = (8, 256) if backend.endswith("8") else (16, 256)
    be = "oracle" if backend == "oracle" else "engine"
    return KatEnv(be, cols, height, rows, n_agents, 0, 1, queue, max_inact, max_steps, reward, library=lib,
                  static_geometry=static, **kw).reset()


# (agents [(x, y, dir, carries_shelf_idx or None)], actions, expected [(x, y)])
MOVES = {
    "down": ([(4, 25, DOWN, None)], [FORWARD], [(4, 26)]),
    "up": ([(4, 25, UP, None)], [FORWARD], [(4, 24)]),
    "left": ([(4, 25, LEFT, None)], [FORWARD], [(3, 25)]),
    "right": ([(4, 25, RIGHT, None)], [FORWARD], [(5, 25)]),
    "wall_up": ([(4, 0, UP, None)], [FORWARD], [(4, 0)]),
    "wall_down": ([(4, 28, DOWN, None)], [FORWARD], [(4, 28)]),
    "wall_left": ([(0, 25, LEFT, None)], [FORWARD], [(0, 25)]),
    "wall_right": ([(9, 25, RIGHT, None)], [FORWARD], [(9, 25)]),
    "swap_unloaded": ([(4, 25, RIGHT, None), (5, 25, LEFT, None)], [FORWARD] * 2, [(4, 25), (5, 25)]),
    "swap_one_loaded": ([(4, 25, RIGHT, 0), (5, 25, LEFT, None)], [FORWARD] * 2, [(4, 25), (5, 25)]),
    "swap_both_loaded": ([(4, 25, RIGHT, 0), (5, 25, LEFT, 1)], [FORWARD] * 2, [(4, 25), (5, 25)]),
    "swap_loaded_into_shelf_column": ([(3, 25, LEFT, 0), (2, 25, RIGHT, None)], [FORWARD] * 2, [(3, 25), (2, 25)]),
    "follow_2": ([(3, 25, RIGHT, None), (4, 25, RIGHT, None)], [FORWARD] * 2, [(4, 25), (5, 25)]),
    "follow_blocked_leader": ([(3, 25, RIGHT, None), (4, 25, RIGHT, None)], [FORWARD, NOOP], [(3, 25), (4, 25)]),
    "chain_beats_side_entry": ([(3, 25, RIGHT, None), (4, 25, RIGHT, None), (5, 26, UP, None)], [FORWARD] * 3,
                               [(4, 25), (5, 25), (5, 26)]),
    "cycle_4": ([(3, 25, RIGHT, None), (4, 25, UP, None), (4, 24, LEFT, None), (3, 24, DOWN, None)], [FORWARD] * 4,
                [(4, 25), (4, 24), (3, 24), (3, 25)]),
    "cycle_4_plus_tail": ([(3, 25, RIGHT, None), (4, 25, UP, None), (4, 24, LEFT, None), (3, 24, DOWN, None),
                           (5, 24, LEFT, None)], [FORWARD] * 5, [(4, 25), (4, 24), (3, 24), (3, 25), (5, 24)]),
    "carry_moves_shelf": ([(4, 25, DOWN, 0)], [FORWARD], [(4, 26)]),
    "loaded_blocked_by_standing_shelf": ([(3, 25, LEFT, 0)], [FORWARD], [(3, 25)]),
    "loaded_chain": ([(3, 25, RIGHT, 0), (4, 25, RIGHT, 1)], [FORWARD] * 2, [(4, 25), (5, 25)]),
}


def _movement_params():
    """(backend, case) pairs; the "-static" backends run the exact-shape build of rware-small-4ag, so cases with more than 4 agents
    are not generated for them (they used to be generated and skipped: a skip in the GPU record reads like a guard that did not run)."""
    out = []
    for case in sorted(MOVES):
        for b in BACKENDS:
            name, marks = (b.values[0], b.marks) if hasattr(b, "values") else (b, ())
            if "-static" in name and len(MOVES[case][0]) > 4:
                continue
            out.append(pytest.param(name, case, marks=marks, id=f"{case}-{name}"))
    return out


@pytest.mark.parametrize("backend,case", _movement_params())
def test_movement_kat(backend, case):
    agents, actions, expect = MOVES[case]
    env = make(backend, len(agents))
    for i, (x, y, d, c) in enumerate(agents):
        a = env.agents[i]
        a.x, a.y, a.dir = x, y, d
        if c is not None:
            a.carrying_shelf = env.shelfs[c]
            env.shelfs[c].x, env.shelfs[c].y = x, y
    env.step(actions)
    for i, (x, y) in enumerate(expect):
        assert (env.agents[i].x, env.agents[i].y) == (x, y), (case, i)
        c = agents[i][3]
        if c is not None:  # a carried shelf travels with its carrier (or stays with it)
            assert (env.shelfs[c].x, env.shelfs[c].y) == (x, y)
    env.close()


@pytest.mark.parametrize("backend", BACKENDS)
def test_moving_under_shelves_until_the_wall(backend):
    env = make(backend, 1)
    a = env.agents[0]
    a.x, a.y, a.dir = 0, 25, RIGHT
    for i in range(10):
        env.step([FORWARD])
        assert (a.x, a.y) == (min(i + 1, 9), 25)
    env.close()


@pytest.mark.parametrize("backend", BACKENDS)
def test_turn_tables(backend):
    # wraplist [UP, RIGHT, DOWN, LEFT] (rware/warehouse.py:119): RIGHT steps forward, LEFT backward
    env = make(backend, 1)
    a = env.agents[0]
    a.x, a.y = 4, 25
    for action, order in ((TURN_RIGHT, [UP, RIGHT, DOWN, LEFT, UP]), (TURN_LEFT, [UP, LEFT, DOWN, RIGHT, UP])):
        a.dir = order[0]
        for want in order[1:]:
            env.step([action])
            assert a.dir == want and (a.x, a.y) == (4, 25)
    env.close()


@pytest.mark.parametrize("backend", BACKENDS)
def test_pickup_carry_and_unload_rules(backend):
    env = make(backend, 1)
    a = env.agents[0]
    a.x, a.y, a.dir = 3, 25, LEFT
    env.step([FORWARD])
    env.step([TOGGLE])                       # picks up the shelf standing at (2, 25)
    shelf = a.carrying_shelf
    assert shelf is not None and (shelf.x, shelf.y) == (2, 25)
    env.step([TURN_LEFT]); env.step([TURN_LEFT]); env.step([FORWARD])
    assert (a.x, a.y, shelf.x, shelf.y) == (3, 25, 3, 25)
    env.step([FORWARD])
    assert (a.x, a.y, shelf.x, shelf.y) == (4, 25, 4, 25)
    env.step([TOGGLE])                       # (4, 25) is a highway: cannot unload
    assert a.carrying_shelf is shelf
    env.step([FORWARD])
    assert (a.x, a.y, shelf.x, shelf.y) == (5, 25, 5, 25)
    env.close()

    env = make(backend, 1)
    a = env.agents[0]
    a.x, a.y, a.dir = 3, 25, LEFT
    env.step([FORWARD]); env.step([TOGGLE])
    shelf = a.carrying_shelf
    env.step([TURN_LEFT]); env.step([TURN_LEFT])
    env.step([TOGGLE])                       # off the highway: unloads
    assert a.carrying_shelf is None
    env.step([FORWARD])
    assert (a.x, a.y, shelf.x, shelf.y) == (3, 25, 2, 25)
    env.close()


def _goal_env(backend, n_agents, reward):
    env = make(backend, n_agents, reward)
    a, s = env.agents[0], env.shelfs[0]
    a.x = s.x = 4
    a.y = s.y = 27
    a.dir = DOWN
    a.carrying_shelf = s
    if n_agents > 1:
        b = env.agents[1]
        b.x, b.y, b.dir = 3, 3, DOWN
    env.request_queue[0] = s
    # keep the queue free of duplicates of shelf 1
    for k in range(1, len(env.request_queue)):
        if env.request_queue[k] is s:
            env.request_queue[k] = next(x for x in env.shelfs if x not in env.request_queue)
    return env


@pytest.mark.parametrize("backend", BACKENDS)
def test_goal_cells_and_delivery_rewards(backend):
    env = _goal_env(backend, 1, GLOBAL)
    assert env.goals == [(4, 28), (5, 28)]
    s = env.shelfs[0]
    _, rew, _, _, _ = env.step([FORWARD])
    assert (env.agents[0].x, env.agents[0].y) == (4, 28)
    assert env.request_queue[0] is not s and s not in env.request_queue and rew == [1.0]
    assert env.agents[0].carrying_shelf is s          # the shelf stays on the agent (:926 quirk aside)
    env.close()

    env = _goal_env(backend, 2, GLOBAL)
    _, rew, _, _, _ = env.step([FORWARD, NOOP])
    assert rew == [1.0, 1.0]
    env.close()

    env = _goal_env(backend, 2, INDIVIDUAL)
    _, rew, _, _, _ = env.step([FORWARD, NOOP])
    assert rew == [1.0, 0.0]
    env.close()

    env = _goal_env(backend, 1, GLOBAL)                # walking away pays nothing
    s = env.shelfs[0]
    for act in (TURN_LEFT, TURN_LEFT):
        assert env.step([act])[1] == [0.0]
    _, rew, _, _, _ = env.step([FORWARD])
    assert (env.agents[0].x, env.agents[0].y) == (4, 26) and env.request_queue[0] is s and rew == [0.0]
    env.close()


@pytest.mark.parametrize("backend", BACKENDS)
def test_two_stage_reward(backend):
    env = _goal_env(backend, 2, TWO_STAGE)
    _, rew, _, _, _ = env.step([FORWARD, NOOP])
    assert rew == [0.5, 0.0] and env.agents[0].has_delivered
    a, s = env.agents[0], env.shelfs[0]
    a.x = s.x = 1
    a.y = s.y = 1
    _, rew, _, _, _ = env.step([TOGGLE, NOOP])         # first off-highway unload after a delivery
    assert rew == [0.5, 0.0]
    for _ in range(2):
        assert env.step([TOGGLE, NOOP])[1] == [0.0, 0.0]
    env.close()


@pytest.mark.parametrize("backend", BACKENDS)
def test_two_deliveries_in_one_step_draw_in_goal_order(backend):
    """Not pinned by the reference's tests (SURVEY.md §4 gaps): both goal cells deliver at once."""
    env = _goal_env(backend, 2, INDIVIDUAL)
    b, s2 = env.agents[1], env.shelfs[1]
    b.x = s2.x = 5
    b.y = s2.y = 27
    b.dir = DOWN
    b.carrying_shelf = s2
    env.request_queue[1] = s2
    for k in range(2, len(env.request_queue)):
        if env.request_queue[k] in (env.shelfs[0], s2):
            env.request_queue[k] = next(x for x in env.shelfs if x not in env.request_queue)
    _, rew, _, _, _ = env.step([FORWARD, FORWARD])
    assert rew == [1.0, 1.0]
    assert env.shelfs[0] not in env.request_queue and s2 not in env.request_queue
    assert len({s.id for s in env.request_queue}) == len(env.request_queue)
    env.close()


@pytest.mark.parametrize("backend", GENERIC_ONLY)
@pytest.mark.parametrize("cols,height,rows", [(1, 2, 2), (3, 8, 1), (3, 8, 3), (5, 3, 2), (7, 4, 4)])
def test_grid_size_formula(backend, cols, height, rows):
    env = make(backend, 1, cols=cols, height=height, rows=rows, queue=1)
    assert tuple(env.grid_size) == ((height + 1) * rows + 2, 3 * cols + 1)   # README / warehouse.py:297-300
    env.close()


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("max_steps", [1, 100, 200])
def test_max_steps_terminates(backend, max_steps):
    if backend == "emu-static" and max_steps > 1:
        max_steps //= 20
    elif backend == "emu" and max_steps > 1:
        max_steps //= 8   # (256 host threads per emulated workgroup: keep the CPU suite short; the GPU runs 100 / 200)
    env = make(backend, 1, max_steps=max_steps)
    for _ in range(max_steps - 1):
        assert env.step([NOOP])[2] is False
    _, _, done, trunc, _ = env.step([NOOP])
    assert done is True and trunc is False
    env.close()


@pytest.mark.parametrize("backend", BACKENDS)
def test_inactivity_limit_and_its_reset_by_a_delivery(backend):
    env = make(backend, 1, max_inact=5)
    for _ in range(4):
        assert env.step([NOOP])[2] is False
    assert env.step([NOOP])[2] is True
    env.close()

    env = _goal_env(backend, 1, GLOBAL)
    env.close()
    env = make(backend, 1, max_inact=5)
    a, s = env.agents[0], env.shelfs[0]
    a.x = s.x = 4
    a.y = s.y = 27
    a.dir = DOWN
    a.carrying_shelf = s
    env.request_queue[0] = s
    for k in range(1, len(env.request_queue)):
        if env.request_queue[k] is s:
            env.request_queue[k] = next(x for x in env.shelfs if x not in env.request_queue)
    for _ in range(3):
        assert env.step([NOOP])[2] is False
    assert env.step([FORWARD])[2] is False and env._cur_inactive_steps == 0   # delivery clears the counter
    for _ in range(4):
        assert env.step([NOOP])[2] is False
    assert env.step([NOOP])[2] is True
    env.close()


@pytest.mark.parametrize("backend", BACKENDS)
def test_equal_length_tie_goes_to_the_lowest_agent_id(backend):
    """The pinned tie-break rule: two single agents contest one empty cell."""
    for first in (0, 1):
        env = make(backend, 2)
        lo, hi = env.agents[first], env.agents[1 - first]
        lo.x, lo.y, lo.dir = 4, 25, RIGHT      # targets (5, 25)
        hi.x, hi.y, hi.dir = 6, 25, LEFT       # targets (5, 25)
        env.step([FORWARD, FORWARD])
        assert (env.agents[0].x, env.agents[0].y) == (5, 25)            # agent id 1 always wins
        other = env.agents[1]
        assert (other.x, other.y) == ((6, 25) if first == 0 else (4, 25))
        env.close()


@pytest.mark.parametrize("backend", BACKENDS)
def test_invalid_action_raises(backend):
    env = make(backend, 1)
    with pytest.raises(ValueError):
        env.step([5])
    env.close()
