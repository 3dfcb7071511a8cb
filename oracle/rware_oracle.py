"""TEST INFRASTRUCTURE — ctypes front-end of the CPU oracle (`rware_oracle.c`).  NOT PRODUCT CODE.

Only tests/, `__graft_entry__.smoke()` and bench.py's `cpu_baseline` leg import this.
State is batched SoA numpy arrays with the same field meaning as the engine's buffers,
so a parity test is `np.array_equal(oracle.<field>, engine.get_state()[<field>])`.

Layout restated from /root/reference/rware/warehouse.py:294-350 (independently of the
product's own `layout.py`, so the two check each other).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "librware_oracle.so")
_SRC = os.path.join(_HERE, "rware_oracle.c")


def build(force: bool = False) -> str:
    alt = os.environ.get("RWARE_ORACLE_SO")  # e.g. the ASAN/UBSAN build made by oracle/sanitize.sh
    if alt:
        return alt
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(_SRC):
        subprocess.check_call(
            ["gcc", "-O2", "-std=c11", "-shared", "-fPIC", "-Wall", "-o", _SO, _SRC]
        )
    return _SO


class _Cfg(C.Structure):
    _fields_ = [
        ("H", C.c_int32), ("W", C.c_int32), ("N", C.c_int32), ("Q", C.c_int32),
        ("R", C.c_int32), ("n_goals", C.c_int32), ("max_inactivity", C.c_int32),
        ("max_steps", C.c_int32), ("reward_type", C.c_int32), ("normalised", C.c_int32),
        ("msg_bits", C.c_int32), ("pad_", C.c_int32),
        ("highways", C.c_void_p), ("goals", C.c_void_p),
    ]


class _State(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "grid", "agent_x", "agent_y", "agent_dir", "agent_carry", "agent_delivered",
        "queue", "steps", "inactive", "rng", "agent_msg")]


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.orc_seed.argtypes = [C.c_uint64, C.c_void_p]
        _lib.orc_seed.restype = None
        _lib.orc_rng_bounded.argtypes = [C.c_void_p, C.c_uint32]
        _lib.orc_rng_bounded.restype = C.c_uint32
        _lib.orc_rng_choice.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        _lib.orc_rng_choice.restype = None
        for f in ("orc_reset", "orc_step", "orc_step_stats", "orc_obs", "orc_recalc_grid", "orc_obs_image"):
            getattr(_lib, f).restype = C.c_int
    return _lib


def layout_from_params(shelf_columns: int, shelf_rows: int, column_height: int):
    """warehouse.py:294-326."""
    assert shelf_columns % 2 == 1, "Only odd number of shelf columns is supported"
    H = (column_height + 1) * shelf_rows + 2
    W = (2 + 1) * shelf_columns + 1
    goals = [(W // 2 - 1, H - 1), (W // 2, H - 1)]
    hw = np.zeros((H, W), dtype=np.uint8)
    for x in range(W):
        for y in range(H):
            hw[y, x] = int(
                x % 3 == 0
                or y % (column_height + 1) == 0
                or y == H - 1
                or (y > H - (column_height + 3) and (x == W // 2 - 1 or x == W // 2))
            )
    return hw, goals


def layout_from_str(layout: str):
    """warehouse.py:328-350."""
    lines = layout.strip().replace(" ", "").split("\n")
    W = len(lines[0])
    assert all(len(l) == W for l in lines), "Layout must be rectangular"
    hw = np.zeros((len(lines), W), dtype=np.uint8)
    goals = []
    for y, line in enumerate(lines):
        for x, ch in enumerate(line):
            assert ch.lower() in "gx."
            if ch.lower() == "g":
                goals.append((x, y))
                hw[y, x] = 1
            elif ch.lower() == ".":
                hw[y, x] = 1
    assert goals, "At least one goal is required"
    return hw, goals


def seed_state(seed: int) -> np.ndarray:
    out = np.zeros(6, dtype=np.uint64)
    lib().orc_seed(C.c_uint64(int(seed)), out.ctypes.data)
    return out


class OracleVecEnv:
    """B independent warehouses stepped by the C oracle (single host thread)."""

    def __init__(self, num_envs, shelf_columns=3, column_height=8, shelf_rows=1, n_agents=2,
                 msg_bits=0, sensor_range=1, request_queue_size=2, max_inactivity_steps=None,
                 max_steps=500, reward_type=1, layout=None, normalised_coordinates=False, observation_type=1,
                 image_observation_layers=None, image_observation_directional=True, **_):
        self.M = int(msg_bits)
        self.observation_type = int(getattr(observation_type, "value", observation_type))  # 1 FLATTENED 2 IMAGE 3 IMAGE_DICT
        assert self.observation_type in (1, 2, 3)
        self.image_layers = tuple(int(getattr(l, "value", l)) for l in (image_observation_layers or self.DEFAULT_IMAGE_LAYERS))
        self.image_directional = bool(image_observation_directional)
        reward_type = int(getattr(reward_type, "value", reward_type))
        self.hw, self.goals = (
            layout_from_str(layout) if layout else layout_from_params(shelf_columns, shelf_rows, column_height)
        )
        self.hw = np.ascontiguousarray(self.hw)
        self._goals = np.ascontiguousarray(np.array(self.goals, dtype=np.int32).reshape(-1))
        self.B, self.N, self.Q, self.R = int(num_envs), int(n_agents), int(request_queue_size), int(sensor_range)
        self.H, self.W = self.hw.shape
        self.S = int((self.hw == 0).sum())
        self.L = 8 + (7 + self.M) * (2 * self.R + 1) ** 2
        self.cfg = _Cfg(self.H, self.W, self.N, self.Q, self.R, len(self.goals),
                        int(max_inactivity_steps or 0), int(max_steps or 0), reward_type,
                        int(bool(normalised_coordinates)), self.M, 0, self.hw.ctypes.data, self._goals.ctypes.data)
        B, N, Q = self.B, self.N, self.Q
        self.grid = np.zeros((B, 2, self.H, self.W), np.int32)
        self.agent_x = np.zeros((B, N), np.int32)
        self.agent_y = np.zeros((B, N), np.int32)
        self.agent_dir = np.zeros((B, N), np.int32)
        self.agent_carry = np.zeros((B, N), np.int32)
        self.agent_delivered = np.zeros((B, N), np.int32)
        self.queue = np.zeros((B, max(Q, 1)), np.int32)[:, :Q]
        self.queue = np.ascontiguousarray(self.queue)
        self.steps = np.zeros(B, np.int32)
        self.inactive = np.zeros(B, np.int32)
        self.rng = np.zeros((B, 6), np.uint64)
        self.agent_msg = np.zeros((B, N), np.int32)
        # event counters, running totals since construction — what the engine's RW_BUF_STAT_* buffers hold (reset() leaves them
        # alone): shelf deliveries (:907-927) and FORWARD requests the reference turned into NOOP (:843-846, :871-876).  Not state:
        # the reference has no such attributes (oracle/ref_runner.py ref_step_events reads the same figures off its objects).
        self.stat_deliveries = np.zeros(B, np.int32)
        self.stat_failed_moves = np.zeros(B, np.int32)
        self._st = None

    FIELDS = ("grid", "agent_x", "agent_y", "agent_dir", "agent_carry", "agent_delivered",
              "queue", "steps", "inactive", "rng")

    def _state(self):
        for f in self.FIELDS:
            a = getattr(self, f)
            assert a.flags.c_contiguous, f
        return _State(*[getattr(self, f).ctypes.data for f in self.FIELDS], self.agent_msg.ctypes.data)

    def get_state(self):
        out = {f: getattr(self, f).copy() for f in self.FIELDS}
        if self.M:
            out["agent_msg"] = self.agent_msg.copy()
        return out

    def set_state(self, **fields):
        for k, v in fields.items():
            getattr(self, k)[...] = v

    def seed(self, seed, mask=None):
        """env i <- SeedSequence(seed + i)  (Gymnasium vector-env convention)."""
        for e in range(self.B):
            if mask is None or mask[e]:
                self.rng[e] = seed_state(int(seed) + e)

    def reset(self, seed=None, mask=None):
        if seed is not None:
            self.seed(seed, mask)
        st = self._state()
        m = None if mask is None else np.ascontiguousarray(np.asarray(mask, np.uint8))
        rc = lib().orc_reset(C.byref(self.cfg), self.B, C.byref(st), None if m is None else m.ctypes.data_as(C.c_void_p))
        assert rc == 0
        return self.obs()

    def step(self, actions, mask=None):
        a = np.ascontiguousarray(np.asarray(actions, dtype=np.int32).reshape(self.B, self.N * (1 + self.M)))
        rew = np.zeros((self.B, self.N), np.float32)
        done = np.zeros(self.B, np.uint8)
        st = self._state()
        m = None if mask is None else np.ascontiguousarray(np.asarray(mask, np.uint8))
        rc = lib().orc_step_stats(C.byref(self.cfg), self.B, C.byref(st), a.ctypes.data_as(C.c_void_p),
                                  rew.ctypes.data_as(C.c_void_p), done.ctypes.data_as(C.c_void_p),
                                  None if m is None else m.ctypes.data_as(C.c_void_p),
                                  self.stat_deliveries.ctypes.data_as(C.c_void_p), self.stat_failed_moves.ctypes.data_as(C.c_void_p))
        if rc == -2:
            raise ValueError("invalid action")
        assert rc == 0, rc
        return rew, done

    def step_autoreset(self, actions, mode="next_step"):
        """One vector-env step with Gymnasium autoreset semantics; returns (obs, rewards, done).

        next_step: an env that reported done on the previous call is reset() now (action ignored,
        reward 0, done False).  same_step: an env that reports done is reset before obs is taken.
        """
        if not hasattr(self, "_prev_done"):
            self._prev_done = np.zeros(self.B, np.uint8)
        if mode == "next_step":
            need = self._prev_done.astype(bool)
            rew, done = self.step(actions, mask=(~need).astype(np.uint8))
            if need.any():
                self.reset(mask=need.astype(np.uint8))
            self._prev_done = done.copy()
            return self.obs(), rew, done
        if mode == "same_step":
            rew, done = self.step(actions)
            if done.any():
                # what Warehouse.step itself returned with done = True (rware/warehouse.py:929-946): Gymnasium's info["final_obs"]
                self.final_obs, self.final_mask = self.obs(), done.astype(bool)
                self.reset(mask=done)
            else:
                self.final_obs, self.final_mask = None, done.astype(bool)
            return self.obs(), rew, done
        rew, done = self.step(actions)
        return self.obs(), rew, done

    def obs(self):
        """Observation in the configured type: FLATTENED (B,N,L); IMAGE (B,N,C,WIN,WIN); IMAGE_DICT (image, features)."""
        if self.observation_type == 2:
            return self.obs_image(self.image_layers, self.image_directional)
        if self.observation_type == 3:
            return self.obs_image(self.image_layers, self.image_directional, with_features=True)
        out = np.zeros((self.B, self.N, self.L), np.float32)
        st = self._state()
        lib().orc_obs(C.byref(self.cfg), self.B, C.byref(st), out.ctypes.data_as(C.c_void_p))
        return out

    DEFAULT_IMAGE_LAYERS = (0, 1, 2, 5, 6)  # SHELVES, REQUESTS, AGENTS, GOALS, ACCESSIBLE (warehouse.py:160-166)
    raise_index_error = True  # mirror the reference's IndexError of the transposed layers

    def obs_image(self, layers=DEFAULT_IMAGE_LAYERS, directional=True, with_features=False):
        """IMAGE observation (B, N, C, WIN, WIN) [+ features (B, N, 6) for IMAGE_DICT]."""
        ly = np.ascontiguousarray(np.asarray([int(getattr(l, "value", l)) for l in layers], np.int32))
        win = 2 * self.R + 1
        out = np.zeros((self.B, self.N, len(ly), win, win), np.float32)
        feat = np.zeros((self.B, self.N, 6), np.float32) if with_features else None
        st = self._state()
        bad = np.zeros(self.B, np.int32)
        rc = lib().orc_obs_image(C.byref(self.cfg), self.B, C.byref(st), ly.ctypes.data_as(C.c_void_p), len(ly),
                                 int(bool(directional)), out.ctypes.data_as(C.c_void_p),
                                 None if feat is None else feat.ctypes.data_as(C.c_void_p),
                                 bad.ctypes.data_as(C.c_void_p))
        assert rc == 0
        self.image_index_error = bad.astype(bool)  # envs where the reference raises IndexError (:552, :558)
        if bad.any() and self.raise_index_error:
            raise IndexError(f"AGENT_DIRECTION / AGENT_LOAD layer: transposed index out of bounds in {int(bad.sum())} env(s) "
                             "(rware/warehouse.py:552,558)")
        return (out, feat) if with_features else out

    def recalc_grid(self, shelf_xy):
        """shelf_xy: (B, S, 2) int32 (x, y) per shelf id, exactly `_recalc_grid` (:749-755)."""
        sx = np.ascontiguousarray(np.asarray(shelf_xy, np.int32).reshape(self.B, -1, 2))
        st = self._state()
        lib().orc_recalc_grid(C.byref(self.cfg), self.B, C.byref(st), sx.ctypes.data_as(C.c_void_p), sx.shape[1])

    def shelf_xy(self):
        """(B, S, 2) positions derived from the shelf layer (inverse of recalc_grid)."""
        out = np.zeros((self.B, self.S, 2), np.int32)
        for e in range(self.B):
            ys, xs = np.nonzero(self.grid[e, 1])
            ids = self.grid[e, 1, ys, xs]
            out[e, ids - 1, 0] = xs
            out[e, ids - 1, 1] = ys
        return out
