"""Batched drop-in for `rware.warehouse.Warehouse` on the step path.

`WarehouseVecEnv` takes the reference constructor's arguments (rware/warehouse.py:146-170)
plus `num_envs`, and exposes the Gymnasium VectorEnv surface:

    obs, info                                   = env.reset(seed=..., options=...)
    obs, rewards, terminated, truncated, info   = env.step(actions)

with obs float32 (B, N, L), rewards float32 (B, N), terminated/truncated bool (B,), info {}.
`obs[:, i, :]` is agent i's FLATTENED observation, i.e. element i of the reference's obs tuple
(:944); `terminated` is the reference's `done` (:935-941); `truncated` is always False (:942).

All simulation runs in the HIP engine (csrc/); this file only marshals arguments.  There is
no CPU path: constructing the env without the built library or without a GPU raises.
"""
from __future__ import annotations

import os

import numpy as np

from . import _capi
from .enums import DEFAULT_IMAGE_LAYERS, Action, ImageLayer, ObservationType, RewardType, enum_value
from .layout import Layout, layout_from_params, layout_from_str, obs_length

try:  # gymnasium is optional (absent in the build image); subclass VectorEnv when present
    import gymnasium as _gym
    from gymnasium.vector import VectorEnv as _VectorEnvBase

    if getattr(_gym, "IS_STANDIN", False):  # never treat the test stand-in as the real package
        raise ImportError
except Exception:  # pragma: no cover - exercised only where gymnasium is missing
    _gym = None

    class _VectorEnvBase:  # minimal duck-typed base
        metadata = {}
        closed = False


def _autoreset_metadata(mode):
    """`metadata["autoreset_mode"]`: a `gymnasium.vector.AutoresetMode` member when the real package is present (what
    Gymnasium >= 1.0 wrappers compare against), the plain string otherwise.  The build image has no gymnasium wheel: this branch
    and `registry.register_gymnasium()` run in CI against a fake of the API (tests/test_gymnasium_boundary.py) and against the real
    package wherever it is installed (tests/test_gymnasium_real.py)."""
    if _gym is not None:
        try:
            from gymnasium.vector import AutoresetMode

            return {"next_step": AutoresetMode.NEXT_STEP, "same_step": AutoresetMode.SAME_STEP,
                    "disabled": AutoresetMode.DISABLED, None: AutoresetMode.DISABLED}[mode]
        except Exception:  # an older gymnasium without AutoresetMode
            pass
    return mode


STATE_FIELDS = ("grid", "agent_x", "agent_y", "agent_dir", "agent_carry", "agent_delivered",
                "queue", "steps", "inactive", "rng")


def global_image_from_state(state, goals, image_layers, pad_to_shape=None):
    """`Warehouse.get_global_image` (rware/warehouse.py:966-1040) from a batched get_state() dict: (B, C, H, W) float32.
    Same quirks: GOALS written as the reference does (its loop variables are swapped twice: the right cells, :1013-1015);
    AGENT_DIRECTION / AGENT_LOAD indexed `[ag.x, ag.y]` on an (H, W) array (:1003-1011) — the mirrored cell, IndexError when
    x >= H or y >= W; REQUESTS from the positions of the queued shelves; padding split as in :1023-1039."""
    grid = np.asarray(state["grid"])
    B, _, H, W = grid.shape
    ax, ay = np.asarray(state["agent_x"]), np.asarray(state["agent_y"])
    rows = np.arange(B)[:, None]
    layers = []
    for lt in image_layers:
        lt = ImageLayer(enum_value(lt))
        layer = np.zeros((B, H, W), np.float32)
        if lt == ImageLayer.SHELVES:
            layer[grid[:, 1] > 0] = 1.0
        elif lt == ImageLayer.REQUESTS:
            q = np.asarray(state["queue"])
            for k in range(q.shape[1]):
                layer[np.asarray(grid[:, 1] == q[:, k, None, None]) & (q[:, k, None, None] > 0)] = 1.0
        elif lt == ImageLayer.AGENTS:
            layer[grid[:, 0] > 0] = 1.0
        elif lt in (ImageLayer.AGENT_DIRECTION, ImageLayer.AGENT_LOAD):
            carry = np.asarray(state["agent_carry"])
            use = np.ones_like(ax, bool) if lt == ImageLayer.AGENT_DIRECTION else carry > 0
            if ((ax >= H) | (ay >= W))[use].any():
                raise IndexError("index out of bounds: AGENT_DIRECTION / AGENT_LOAD are written at [ag.x, ag.y] (rware/warehouse.py:1006,1011)")
            val = (np.asarray(state["agent_dir"]) + 1).astype(np.float32) if lt == ImageLayer.AGENT_DIRECTION else np.ones_like(ax, np.float32)
            for i in range(ax.shape[1]):  # agent order: a later agent overwrites an earlier one on the same (mirrored) cell
                m = use[:, i]
                layer[rows[m, 0], ax[m, i], ay[m, i]] = val[m, i]
        elif lt == ImageLayer.GOALS:
            for gx, gy in goals:
                layer[:, gy, gx] = 1.0
        elif lt == ImageLayer.ACCESSIBLE:
            layer[:] = 1.0
            layer[rows, ay, ax] = 0.0
        layers.append(layer)
    img = np.stack(layers, axis=1)
    if pad_to_shape is not None:
        pad = [p - g for p, g in zip(pad_to_shape, img.shape[1:])]
        assert all(d >= 0 for d in pad)
        img = np.pad(img, ((0, 0),) + tuple((d // 2, d // 2 if d % 2 == 0 else d // 2 + 1) for d in pad), mode="constant", constant_values=0)
    return img


class _Space:
    """Shape/dtype descriptor used when gymnasium is not installed."""

    def __init__(self, shape, dtype, n=None):
        self.shape, self.dtype, self.n = tuple(shape), np.dtype(dtype), n

    def __repr__(self):
        return f"Space(shape={self.shape}, dtype={self.dtype}, n={self.n})"


class _CapturedLoop:
    """What WarehouseVecEnv.capture_loop returns: `steps` (policy, step) rounds in one HIP graph."""

    def __init__(self, graph, stream, keep, steps, engine=None, bridge=False):
        self.graph, self.stream, self._keep, self.steps, self._engine, self._bridge = graph, stream, keep, steps, engine, bridge

    def replay(self):
        import torch

        if self._bridge:
            # the env was built on the default stream and moved to a capture stream of its own: order the replay behind what the
            # caller enqueued so far (a learner still reading the last observations) and the caller's next ops behind the replay
            cur = torch.cuda.current_stream(self.stream.device)
            self.stream.wait_stream(cur)
            with torch.cuda.stream(self.stream):
                self.graph.replay()
            cur.wait_stream(self.stream)
        else:
            with torch.cuda.stream(self.stream):
                self.graph.replay()
        if self._engine is not None:  # the replayed steps ran without host code: get_state() / set_state() must not trust
            self._engine.mark_views_stale()  # the engine's "views are current" flags (grid, agent_* arrays)


class WarehouseVecEnv(_VectorEnvBase):
    metadata = {"render_modes": [], "autoreset_mode": "next_step"}

    def __init__(self, num_envs: int, shelf_columns: int = 3, column_height: int = 8, shelf_rows: int = 1,
                 n_agents: int = 2, msg_bits: int = 0, sensor_range: int = 1, request_queue_size: int = 2,
                 max_inactivity_steps=None, max_steps=500, reward_type=RewardType.INDIVIDUAL,
                 layout: str | None = None, observation_type=ObservationType.FLATTENED,
                 image_observation_layers=None, image_observation_directional: bool = True,
                 normalised_coordinates: bool = False, render_mode=None, *,
                 autoreset_mode: str = "next_step", devices=None, output: str = "numpy",
                 envs_per_workgroup: int = 0, threads_per_workgroup: int = 0, library: str | None = None,
                 obs_stores: str | None = None, jit=None, pipe=None, stats: bool = False, wave_priority=None):
        if not 0 <= int(msg_bits) <= 16:
            raise ValueError("msg_bits must be in 0..16")
        self.msg_bits = int(msg_bits)
        self.observation_type = ObservationType(enum_value(observation_type))
        # DICT (rware/warehouse.py:676-720): the engine produces the FLATTENED vector — spaces.flatten() of exactly that
        # nested dict (:432-443, :505-522) — and the host hands out the batched dict as views / casts of it.
        self._dict_obs = self.observation_type == ObservationType.DICT
        engine_obs_type = ObservationType.FLATTENED if self._dict_obs else self.observation_type
        layers = tuple(ImageLayer(enum_value(l)) for l in (image_observation_layers or DEFAULT_IMAGE_LAYERS))
        # AGENT_DIRECTION / AGENT_LOAD are written with transposed indices by the reference (rware/warehouse.py:552,558):
        # reproduced as is, including its IndexError once an agent stands at x >= grid height or y >= grid width
        image = self.observation_type in (ObservationType.IMAGE, ObservationType.IMAGE_DICT)
        self._index_layers = image and any(
            l in (ImageLayer.AGENT_DIRECTION, ImageLayer.AGENT_LOAD) for l in layers)
        self.image_observation_layers = layers
        self.image_observation_directional = bool(image_observation_directional)
        if output not in ("numpy", "torch"):
            raise ValueError("output must be 'numpy' or 'torch'")
        self.layout: Layout = layout_from_str(layout) if layout else layout_from_params(shelf_columns, shelf_rows, column_height)
        self.num_envs = int(num_envs)
        self.n_agents = int(n_agents)
        self.sensor_range = int(sensor_range)
        self.request_queue_size = int(request_queue_size)
        self.max_inactivity_steps = max_inactivity_steps
        self.max_steps = max_steps
        self.reward_type = RewardType(enum_value(reward_type))
        self.normalised_coordinates = bool(normalised_coordinates)
        self.grid_size = self.layout.grid_size
        self.highways = self.layout.highways
        self.goals = list(self.layout.goals)
        self.autoreset_mode = autoreset_mode
        self.metadata = dict(self.metadata, autoreset_mode=_autoreset_metadata(autoreset_mode))
        self.output = output
        self.obs_length = obs_length(self.sensor_range, self.msg_bits)
        self._seeded = False
        self.closed = False

        devices = [0] if devices is None else list(devices)
        # output="torch" with several devices: every result is a tuple with one zero-copy tensor per device (shard order,
        # env ranges in `shard_bounds`); step() takes one CUDA tensor per device (or one host array) and launches on every
        # device before anything waits ("8 async launches ... default to device-resident tensors")
        base, rem = divmod(self.num_envs, len(devices))
        self._bounds, lo = [], 0
        for d in range(len(devices)):
            n = base + (1 if d < rem else 0)
            self._bounds.append((lo, lo + n))
            lo += n
        self._torch = None
        self._stream_handles = []  # (output="torch": the stream every engine enqueues on — torch's current stream at construction)
        self.engines = []
        for dev, (lo, hi) in zip(devices, self._bounds):
            if hi == lo:
                continue
            stream = None
            if output == "torch":
                import torch

                self._torch = torch
                # enqueue on torch's current stream of that device; its handle is 0 (NULL) for the default stream,
                # which is why the engine is told to take the handle literally (RW_STREAM_USE_GIVEN): launches are
                # then ordered with the policy / learner ops around them and need no sync
                stream = torch.cuda.current_stream(dev).cuda_stream
                self._stream_handles.append(int(stream))
            self.engines.append(_capi.Engine(
                num_envs=hi - lo, layout=self.layout, n_agents=self.n_agents, sensor_range=self.sensor_range,
                request_queue_size=self.request_queue_size, max_inactivity_steps=max_inactivity_steps,
                max_steps=max_steps, reward_type=self.reward_type.value,
                normalised_coordinates=normalised_coordinates, autoreset_mode=autoreset_mode, device_id=dev,
                envs_per_workgroup=envs_per_workgroup, threads_per_workgroup=threads_per_workgroup,
                stream=stream, use_given_stream=(output == "torch"), library=library,
                observation_type=engine_obs_type.value,
                image_layers=[l.value for l in layers] if image else (),
                image_directional=image_observation_directional, msg_bits=self.msg_bits,
                # None / "auto": the engine's measured rule; "cached": observation lines stay in the cache hierarchy for a
                # learner that reads them right behind the step; "stream": non-temporal stores (rw_stream_flags)
                obs_stores=obs_stores,
                # None / "auto": shapes without an ahead-of-time exact-shape kernel are specialised at construction (hipRTC, disk
                # cache) when the batch has >= 4096 envs; False / "off": never; True / "force": always
                jit=jit,
                # None / "auto": the engine's measured rule; False / "off", True / "on": the chunk-pipelined persistent per-step
                # kernel never / wherever the library has a build for the shape (rw_stream_flags RW_PIPE_*)
                pipe=pipe,
                # True: the engine keeps per-env event counters (RW_STATS_ON) — see event_counters()
                stats=stats,
                # None / "auto": the engine's measured rule (rw_info.wave_priority); False / True: the launches never / always run the chain
                # in front of their first observation store at raised wavefront priority (RW_PRIO_OFF / RW_PRIO_ON) — a scheduling hint
                wave_priority=wave_priority))
        self._bounds = [b for b in self._bounds if b[1] > b[0]]
        self.shard_bounds = list(self._bounds)  # env range [lo, hi) of every engine / device, in order
        self.devices = devices[: len(self.engines)]
        self.n_shelves = self.engines[0].S
        self._make_spaces()
        self._tviews = {}
        self._live_actions = None
        self._pool = None
        self._multi = None
        # step()'s fast path: (tensor type, dtype, shape, device, bound C function, engine handle, cached result tuple); only for
        # output="torch" envs whose results are the same zero-copy views every step and whose observation needs no host-side check
        self._has_final_obs = autoreset_mode == "same_step"
        self._fast = None
        self._fast_ok = output == "torch" and not self._dict_obs and len(devices) == 1

    # ------------------------------------------------------------------------------- spaces
    def _make_spaces(self):
        n, l, b = self.n_agents, self.obs_length, self.num_envs
        if self.observation_type in (ObservationType.IMAGE, ObservationType.IMAGE_DICT):
            win = 2 * self.sensor_range + 1
            shape = (len(self.image_observation_layers), win, win)
            self.single_observation_space = tuple(_Space(shape, np.float32) for _ in range(n))
            self.single_action_space = tuple(_Space((), np.int64, n=len(Action)) for _ in range(n))
            self.observation_space = _Space((b, n) + shape, np.float32)
            self.action_space = _Space((b, n), np.int64, n=len(Action))
            return
        if _gym is not None:
            sp = _gym.spaces
            sa_obs = sp.Box(low=-float("inf"), high=float("inf"), shape=(l,), dtype=np.float32)
            self.single_observation_space = sp.Tuple(tuple(n * [sa_obs]))      # rware/warehouse.py:505-522
            sa_act = sp.Discrete(len(Action)) if not self.msg_bits else sp.MultiDiscrete([len(Action)] + self.msg_bits * [2])
            self.single_action_space = sp.Tuple(tuple(n * [sa_act]))  # :255-260
            self.observation_space = sp.Box(-float("inf"), float("inf"), shape=(b, n, l), dtype=np.float32)
            nvec = np.full((b, n), len(Action)) if not self.msg_bits else np.tile([len(Action)] + self.msg_bits * [2], (b, n, 1))
            self.action_space = sp.MultiDiscrete(nvec)
        else:
            self.single_observation_space = tuple(_Space((l,), np.float32) for _ in range(n))
            self.single_action_space = tuple(_Space((), np.int64, n=len(Action)) for _ in range(n))
            self.observation_space = _Space((b, n, l), np.float32)
            self.action_space = _Space((b, n), np.int64, n=len(Action))

    # ------------------------------------------------------------------------------- hot path
    def reset(self, *, seed=None, options=None, mask=None):
        """Warehouse.reset (:757-802) for every env (or those in `mask`).  `seed` int -> env i gets
        seed + i (Gymnasium vector convention); a sequence gives per-env seeds; None continues each
        env's own PCG64 stream (first ever reset: a fresh OS-entropy base seed)."""
        seeds = None
        if seed is None and not self._seeded:
            seed = int.from_bytes(os.urandom(7), "little")
        if seed is not None:
            if np.isscalar(seed):
                if int(seed) < 0:
                    raise ValueError(f"Seed must be a non-negative integer, got {seed!r}")
                seeds = (np.uint64(int(seed)) + np.arange(self.num_envs, dtype=np.uint64)).astype(np.uint64)
            else:
                seeds = np.asarray(seed, dtype=np.uint64).reshape(self.num_envs)
            self._seeded = True
        m = None if mask is None else np.asarray(mask).astype(np.uint8).reshape(self.num_envs)
        for eng, (lo, hi) in zip(self.engines, self._bounds):
            eng.reset(None if seeds is None else seeds[lo:hi], None if m is None else m[lo:hi])
        return self._observations(), {}

    def step(self, actions):
        """Warehouse.step (:804-946) for every env; actions (B, N) ints in 0..4 or Action members.

        With output="torch" all four results are zero-copy views of engine memory — the SAME tensor objects every call,
        overwritten in place by the next step (obs, rewards, terminated; truncated stays False).  A loop that keeps them
        across steps (`dones.append(terminated)`) must `.clone()` them; output="numpy" returns fresh host arrays."""
        fast = self._fast
        if fast is not None and type(actions) is fast[0] and actions.dtype is fast[1] and actions.is_contiguous() \
                and actions.shape == fast[2] and actions.device == fast[3]:
            # the closed loop's hot call: a contiguous int32 CUDA tensor of the right shape on the env's device, single engine —
            # one pre-bound C call and the cached result tuple (the checks of step_async / step_wait, done once)
            self._live_actions = actions
            rc = fast[4](fast[5], actions.data_ptr())
            if rc:
                self.engines[0]._check(rc)
            return fast[6]
        self.step_async(actions)
        out = self.step_wait()
        if fast is None and self._fast_ok and self._torch is not None and isinstance(actions, self._torch.Tensor) and actions.is_cuda \
                and actions.dtype == self._torch.int32 and actions.is_contiguous() and len(self.engines) == 1 \
                and tuple(actions.shape) in ((self.num_envs, self.n_agents), (self.num_envs, self.n_agents, 1 + self.msg_bits)):
            eng = self.engines[0]
            self._fast = (type(actions), actions.dtype, actions.shape, actions.device, eng.lib.rw_step_device, eng._h, out)
        return out

    def step_async(self, actions):
        t = self._torch
        if t is not None and isinstance(actions, (list, tuple)) and actions and all(isinstance(a, t.Tensor) for a in actions):
            # one CUDA tensor per device (the shards of a multi-device env): every launch is enqueued before anything waits
            if len(actions) != len(self.engines):
                raise ValueError(f"expected {len(self.engines)} per-device action tensors, got {len(actions)}")
            live = [self._device_actions(a, eng.B, d) for d, (eng, a) in enumerate(zip(self.engines, actions))]
            if len(live) > 1:  # ONE C call: every engine but the first has a launcher thread of its own (rw_multi)
                if self._multi is None:
                    self._multi = _capi.MultiEngine(self.engines)
                self._multi.step_device([a.data_ptr() for a in live])
            else:
                self.engines[0].step_device(live[0].data_ptr())
            self._live_actions = live
            return
        if t is not None and isinstance(actions, t.Tensor) and actions.is_cuda:
            if len(self.engines) != 1:
                raise ValueError("a multi-device env takes one CUDA action tensor per device (a list), or a host array")
            actions = self._device_actions(actions, self.num_envs, 0)
            # Kept alive until the NEXT step call replaces it.  That is enough for torch's caching allocator, which is
            # stream-ordered: memory freed on this stream is only handed to work enqueued later on the same stream, i.e.
            # behind the step kernel that reads it (a tensor allocated on another stream needs record_stream by its owner).
            self._live_actions = actions
            self.engines[0].step_device(actions.data_ptr())
            return
        a = np.asarray(actions)
        if a.dtype == object:
            a = np.vectorize(enum_value, otypes=[np.int64])(a)
        am = 1 + self.msg_bits  # per-agent action: [Action, message bits...] (rware/warehouse.py:255-259)
        if a.size != self.num_envs * self.n_agents * am:
            raise AssertionError(f"expected {self.num_envs}x{self.n_agents}" + (f"x{am}" if self.msg_bits else "")
                                 + f" actions, got shape {a.shape}")  # :807
        a = a.reshape(self.num_envs, self.n_agents, am)
        if a.size and (a[..., 0].min() < 0 or a[..., 0].max() > 4):
            bad = a[..., 0][(a[..., 0] < 0) | (a[..., 0] > 4)].flat[0]
            raise ValueError(f"{bad} is not a valid Action")  # Action(action) at :811/:814
        if self.msg_bits and (a[..., 1:].min() < 0 or a[..., 1:].max() > 1):
            raise ValueError("message bits must be 0 or 1 (MultiDiscrete([5, 2, ...]))")
        a = a.astype(np.int32, copy=False)
        for eng, (lo, hi) in zip(self.engines, self._bounds):
            eng.step_host(a[lo:hi])

    def _device_actions(self, actions, n_envs, d):
        t = self._torch
        if not actions.is_cuda or actions.device.index != self.devices[d]:
            raise ValueError(f"actions for shard {d} must live on cuda:{self.devices[d]}")
        if actions.numel() != n_envs * self.n_agents * (1 + self.msg_bits):
            raise ValueError("device actions must hold B*N*(1+msg_bits) elements")
        if actions.dtype in (t.int64, t.int16, t.int8, t.uint8) or not actions.is_contiguous():
            # what a policy usually hands over (argmax / Categorical.sample() are int64): one small cast on the same
            # stream, ordered before the step like any other torch op
            actions = actions.to(t.int32).contiguous()
        if actions.dtype != t.int32:
            raise ValueError("device actions must be an integer tensor")
        return actions

    def step_wait(self):
        if self.output == "torch":
            if len(self.engines) > 1:  # one tuple entry per device, in shard order; nothing waited for
                vs = [self._torch_views(d) for d in range(len(self.engines))]
                fin = [self._final_info(v, d) for d, v in enumerate(vs)]
                info = {k: tuple(f[k] for f in fin) for k in fin[0]}  # (one entry per device, like the results)
                return (tuple(self._obs_of(v) for v in vs), tuple(v["rewards"] for v in vs),
                        tuple(v["terminated_bool"] for v in vs), tuple(v["truncated_bool"] for v in vs), info)
            v = self._torch_views()
            # the flag buffers are uint8 0/1: reinterpreted as bool, not cast (a cast is a torch kernel per flag per step
            # around a ~7 us step kernel)
            return self._observations(), v["rewards"], v["terminated_bool"], v["truncated_bool"], self._final_info(v)
        # host arrays: the whole return tuple in one round trip per device (`truncated` is always False, :942 — nothing to read)
        if self._index_layers:
            self.sync()  # IndexError where the reference's _make_img_obs raises it
        want_f = self.observation_type == ObservationType.IMAGE_DICT
        if len(self.engines) > 1:
            if self._pool is None:
                from concurrent.futures import ThreadPoolExecutor

                self._pool = ThreadPoolExecutor(len(self.engines))
            parts = list(self._pool.map(lambda eng: eng.read_outputs(want_f), self.engines))
            obs, rew, term = (np.concatenate([p[k] for p in parts], axis=0) for k in range(3))
            feat = np.concatenate([p[3] for p in parts], axis=0) if want_f else None
        else:
            obs, rew, term, feat = self.engines[0].read_outputs(want_f)
        info = {}
        if self._has_final_obs and term.any():  # (rare: the extra copy is made only on a step that ended an episode)
            info = {"final_obs": self._gather("final_obs"), "_final_obs": term.view(np.bool_).copy()}
            if self._dict_obs:
                info["final_obs"] = self.dict_from_flat(info["final_obs"])
            elif want_f:  # IMAGE_DICT: the terminal observation is a dict like every other one (rware/warehouse.py:739-742)
                info["final_obs"] = {"image": info["final_obs"], "features": self._gather("final_features")}
        if self._dict_obs:
            obs = self.dict_from_flat(obs)
        elif want_f:
            obs = {"image": obs, "features": feat}
        return obs, rew, term.view(np.bool_), np.zeros(self.num_envs, np.bool_), info

    def rollout(self, actions, want_obs=True):
        """Open-loop rollout: `actions` (T, B, N) -> (obs (T,B,N,L), rewards (T,B,N), terminated (T,B)).
        One fused kernel launch per shard (`rw_step_many_device`): the env chunk stays in LDS across
        the T steps.  Bit-identical to T calls of step() with the same actions."""
        t = self._torch
        if t is not None and isinstance(actions, t.Tensor) and actions.is_cuda:
            return self._rollout_device(actions, want_obs)
        a = np.asarray(actions)
        am = 1 + self.msg_bits
        a = a.reshape(a.shape[0], self.num_envs, self.n_agents, am) if a.size and a.size % (self.num_envs * self.n_agents * am) == 0 else a
        if a.ndim != 4 or a.shape[1:] != (self.num_envs, self.n_agents, am):
            raise AssertionError(f"expected (T, {self.num_envs}, {self.n_agents}" + (f", {am}" if self.msg_bits else "") + f") actions, got {a.shape}")
        if a.size and (a[..., 0].min() < 0 or a[..., 0].max() > 4 or (self.msg_bits and (a[..., 1:].min() < 0 or a[..., 1:].max() > 1))):
            raise ValueError("invalid Action in tape")
        a = a.reshape(a.shape[0], self.num_envs, self.n_agents * am)
        parts = [eng.rollout_host(a[:, lo:hi], want_obs) for eng, (lo, hi) in zip(self.engines, self._bounds)]
        cat = lambda k: parts[0][k] if len(parts) == 1 else np.concatenate([p[k] for p in parts], axis=1)
        return (cat(0) if want_obs else None), cat(1), cat(2).astype(bool)

    def _rollout_device(self, actions, want_obs):
        """rollout() with a CUDA action tape (T, B, N[, 1+M]) of an output="torch" env: the tapes come back as torch tensors
        on the same device — obs (T, B, N, L) float32, rewards (T, B, N) float32, terminated (T, B) bool — written by the one
        fused launch on torch's current stream; nothing is copied, nothing is waited for, no device allocation outside torch's
        caching allocator.  Out-of-range actions run as NOOP and raise at the next sync() (as for step())."""
        t = self._torch
        if len(self.engines) != 1:
            raise ValueError("device rollouts are per device: one env per GPU")
        eng, am = self.engines[0], 1 + self.msg_bits
        per_step = self.num_envs * self.n_agents * am
        if actions.numel() == 0 or actions.numel() % per_step:
            raise AssertionError(f"expected (T, {self.num_envs}, {self.n_agents}" + (f", {am}" if self.msg_bits else "") + f") actions, got {tuple(actions.shape)}")
        T = actions.numel() // per_step
        if actions.dtype != t.int32 or not actions.is_contiguous():
            actions = actions.to(t.int32).contiguous()
        dev = actions.device
        obs = t.empty((T,) + tuple(eng.shapes["obs"]), dtype=t.float32, device=dev) if want_obs else None
        rew = t.empty((T, self.num_envs, self.n_agents), dtype=t.float32, device=dev)
        term = t.empty((T, self.num_envs), dtype=t.uint8, device=dev)
        self._live_actions = actions
        eng.step_many_device(actions.data_ptr(), T, obs.data_ptr() if want_obs else 0, rew.data_ptr(), term.data_ptr())
        return obs, rew, term.view(t.bool)

    def capture_loop(self, policy, steps: int = 1, warmup: int = 2):
        """`steps` rounds of  `actions = policy(obs, rewards, terminated); env.step(actions)`  captured in ONE HIP graph.

        A step is a single ~6 us kernel; a torch policy in front of it is several small kernels (a one-layer policy: 37 us
        per round issued eagerly, 32 us replayed from a graph — profiles/r03_learner_probe.txt).  Replaying policy + step
        from a graph removes the host's share of that.  `policy` gets the env's zero-copy output tensors (the same objects every round: they are overwritten in
        place by the step) and returns an integer CUDA tensor of (B, N) actions; it must be capturable (no host syncs, no
        data-dependent shapes).  A HIP graph cannot be captured on the legacy default stream: an env that was built there
        (the usual case) is moved to a stream of its own for good (rw_set_stream) — replay() then orders that stream behind the
        caller's current stream and the caller's next ops behind the replay (two event waits per replay, no host sync), and
        eager env.step() calls keep working (they enqueue on the env's new stream: order them with `loop.stream` yourself).
        `policy` is called `warmup` times eagerly first (outputs discarded, the env does not step) so that lazy library
        initialisation happens outside the capture.
        Returns an object with `.replay()` (enqueue the captured rounds once more), `.graph` and `.stream`."""
        t = self._torch
        if t is None or len(self.engines) != 1:
            raise ValueError('capture_loop needs output="torch" and a single device')
        bridge = False
        if not self._stream_handles or not self._stream_handles[0]:
            own = t.cuda.Stream(device=self.devices[0])
            own.wait_stream(t.cuda.current_stream(self.devices[0]))     # everything enqueued so far comes first
            self.engines[0].set_stream(own.cuda_stream)
            self._stream_handles = [int(own.cuda_stream)]
            self._own_stream = own                                      # (keeps the stream alive as long as the env)
            self._fast = None
            bridge = True
        dev = self.devices[0]
        s = t.cuda.ExternalStream(self._stream_handles[0], device=f"cuda:{dev}")
        v = self._torch_views()
        obs = self._obs_of(v)
        keep = []
        with t.cuda.stream(s):
            for _ in range(max(0, int(warmup))):
                policy(obs, v["rewards"], v["terminated_bool"])
            s.synchronize()
            g = t.cuda.CUDAGraph()
            with t.cuda.graph(g, stream=s):
                for _ in range(int(steps)):
                    a = self._device_actions(policy(obs, v["rewards"], v["terminated_bool"]), self.num_envs, 0)
                    keep.append(a)  # (allocated from the graph's private pool: alive as long as the graph is)
                    self.engines[0].step_device(a.data_ptr())
        return _CapturedLoop(g, s, keep, int(steps), self.engines[0], bridge)

    def snapshot(self):
        """Checkpoint the batched state on the device (grid, agents, queue, counters, RNG streams).
        Returns an opaque token for restore(); free it with free_snapshot()."""
        return [eng.snapshot() for eng in self.engines]

    def restore(self, token):
        """Roll every env back to a snapshot(); the engine then continues bit-identically."""
        for eng, h in zip(self.engines, token):
            eng.restore(h)
        return self._observations()

    def free_snapshot(self, token):
        for eng, h in zip(self.engines, token):
            eng.free_snapshot(h)

    def trim(self):
        """Releases the device tapes rollout() with host arrays keeps between calls (one rollout(T=100) of 16384 small-4ag
        envs with observations holds ~1.9 GB per shard until then; they are re-allocated on the next such call)."""
        for eng in self.engines:
            eng.release_arena()

    def sync(self):
        """Wait for enqueued work; raises ValueError if a device-side action was out of range."""
        for eng in self.engines:
            eng.sync()

    # ------------------------------------------------------------------------------- buffers
    def _gather(self, name):
        if len(self.engines) > 1:  # one reader thread per device: ctypes drops the GIL, the PCIe copies overlap
            if self._pool is None:
                from concurrent.futures import ThreadPoolExecutor

                self._pool = ThreadPoolExecutor(len(self.engines))
            parts = list(self._pool.map(lambda eng: eng.read(name), self.engines))
        else:
            parts = [eng.read(name) for eng in self.engines]
        return parts[0] if len(parts) == 1 else np.concatenate(parts, axis=1 if name == "rng" else 0)

    def _observations(self):
        """FLATTENED: (B, N, L).  IMAGE: (B, N, C, 2r+1, 2r+1).  IMAGE_DICT: {"image": ..., "features": (B, N, 6)}
        (the batched form of the reference's per-agent dicts, rware/warehouse.py:739-742)."""
        if self.output == "torch":
            if self._dict_obs:
                raise NotImplementedError("DICT observations are host-side views of the FLATTENED batch: use output='numpy' "
                                          "(or FLATTENED with output='torch')")
            if len(self.engines) > 1:
                return tuple(self._obs_of(self._torch_views(d)) for d in range(len(self.engines)))
            return self._obs_of(self._torch_views())
        if self._index_layers:
            self.sync()  # IndexError where the reference's _make_img_obs raises it (device tensors: at sync())
        obs = self._gather("obs")
        if self._dict_obs:
            return self.dict_from_flat(obs)
        if self.observation_type == ObservationType.IMAGE_DICT:
            return {"image": obs, "features": self._gather("features")}
        return obs

    def dict_from_flat(self, flat):
        """The DICT observation (rware/warehouse.py:676-720) of every agent, batched: same nesting and keys as the
        reference's per-agent dict, every leaf an array with leading (B, N) — `location` (B,N,2) int32, the
        MultiBinary(1) fields (B,N,1), `direction` (B,N), `local_message` (B,N,M) or None, `sensors` a tuple of
        (2r+1)^2 dicts in the reference's row-major window order.  `flat` is the FLATTENED batch (B, N, L)."""
        M, cells = self.msg_bits, (2 * self.sensor_range + 1) ** 2
        i8 = lambda a: a.astype(np.int64)
        out = {"self": {
            "location": flat[..., 0:2].astype(np.int32),  # np.array([x, y], dtype=np.int32) — truncates normalised coordinates too (:685)
            "carrying_shelf": i8(flat[..., 2:3]),
            "direction": i8(flat[..., 3:7].argmax(-1)),
            "on_highway": i8(flat[..., 7:8]),
        }}
        sensors = []
        for c in range(cells):
            o = 8 + (7 + M) * c
            sensors.append({
                "has_agent": i8(flat[..., o:o + 1]),
                "direction": i8(flat[..., o + 1:o + 5].argmax(-1)),  # 0 for an empty cell (:697)
                "local_message": i8(flat[..., o + 5:o + 5 + M]) if M else None,
                "has_shelf": i8(flat[..., o + 5 + M:o + 6 + M]),
                "shelf_requested": i8(flat[..., o + 6 + M:o + 7 + M]),
            })
        out["sensors"] = tuple(sensors)
        return out

    @staticmethod
    def unbatch_dict_obs(obs, b):
        """Env b of a batched DICT observation in the reference's own form: a tuple of N per-agent dicts with
        lists / ints / arrays exactly as `_get_default_obs` builds them (:676-720)."""
        n = obs["self"]["direction"].shape[1]

        def agent(i):
            s = obs["self"]
            return {
                "self": {"location": s["location"][b, i].copy(), "carrying_shelf": [int(s["carrying_shelf"][b, i, 0])],
                         "direction": int(s["direction"][b, i]), "on_highway": [int(s["on_highway"][b, i, 0])]},
                "sensors": tuple({
                    "has_agent": [int(c["has_agent"][b, i, 0])], "direction": int(c["direction"][b, i]),
                    "local_message": None if c["local_message"] is None else [int(v) for v in c["local_message"][b, i]],
                    "has_shelf": [int(c["has_shelf"][b, i, 0])], "shelf_requested": [int(c["shelf_requested"][b, i, 0])],
                } for c in obs["sensors"]),
            }
        return tuple(agent(i) for i in range(n))

    def _final_info(self, v, d=0):
        """SAME_STEP autoreset: the step that ends an episode returns the RESET observation; what the reference's step() returned
        with done = True (rware/warehouse.py:929-946) is kept as info["final_obs"], valid in the rows where info["_final_obs"]
        (== terminated) is set — Gymnasium >= 1.0's vector-env convention.  Zero-copy views (torch output)."""
        if not self._has_final_obs:
            return {}
        if "final_obs" not in v:
            t, eng = self._torch, self.engines[d]
            v["final_obs"] = t.as_tensor(eng.device_array("final_obs"), device=v["obs"].device)
            if self.observation_type == ObservationType.IMAGE_DICT:
                v["final_obs"] = {"image": v["final_obs"], "features": t.as_tensor(eng.device_array("final_features"), device=v["obs"].device)}
        return {"final_obs": v["final_obs"], "_final_obs": v["terminated_bool"]}

    def _obs_of(self, v):
        return {"image": v["obs"], "features": v["features"]} if self.observation_type == ObservationType.IMAGE_DICT else v["obs"]

    def _torch_views(self, d=0):
        v = self._tviews.get(d)
        if v is None:
            t, eng, dev = self._torch, self.engines[d], self.devices[d]
            v = {k: t.as_tensor(eng.device_array(k), device=f"cuda:{dev}")
                 for k in ("obs", "rewards", "terminated", "truncated", "features")}
            v["terminated_bool"] = v["terminated"].view(t.bool)  # uint8 0/1 reinterpreted: no kernel, no copy
            v["truncated_bool"] = v["truncated"].view(t.bool)
            self._tviews[d] = v
        return v

    def event_counters(self):
        """{"deliveries": (B,), "failed_moves": (B,)} int32 — running totals per env since construction (`stats=True` only; the
        engine never resets them: an episode's figure is the difference between two reads).  deliveries: requested shelves brought
        to a goal (rware/warehouse.py:907-917); failed_moves: FORWARD requests the step turned into NOOP — the shelf-block cancel
        (:836-846) and the movers that lose the collision resolution (:871-876).  The reference keeps no such figures: its `info`
        is {} (:746-747), and so is this env's.  output="torch": zero-copy device tensors (a tuple per device when sharded)."""
        if not self.engines[0].stats:
            raise RuntimeError("event counters are off: construct the env with stats=True")
        names = {"deliveries": "stat_deliveries", "failed_moves": "stat_failed_moves"}
        if self.output == "torch":
            per_dev = [{k: self._torch.as_tensor(eng.device_array(n), device=f"cuda:{dev}") for k, n in names.items()}
                       for eng, dev in zip(self.engines, self.devices)]
            return per_dev[0] if len(per_dev) == 1 else tuple(per_dev)
        return {k: self._gather(n) for k, n in names.items()}

    def device_tensor(self, name):
        """Zero-copy torch view of any engine buffer (single-device envs)."""
        import torch

        return torch.as_tensor(self.engines[0].device_array(name), device=f"cuda:{self.devices[0]}")

    # ------------------------------------------------------------------------------- state
    def get_state(self) -> dict:
        """Batched SoA state: grid (B,2,H,W), agent_* (B,N), queue (B,Q), steps/inactive (B,), rng (B,6)."""
        out = {k: self._gather(k) for k in STATE_FIELDS + (("agent_msg",) if self.msg_bits else ())}
        out["rng"] = np.ascontiguousarray(out["rng"].T)
        return out

    def set_state(self, refresh_obs: bool = True, **fields):
        # `grid` first: later coordinate writes then re-mark the derived int32 view stale (layer 0 follows agent_x / agent_y)
        for k, v in sorted(fields.items(), key=lambda kv: kv[0] != "grid"):
            if k not in STATE_FIELDS and k not in ("need_reset", "agent_msg", "stat_deliveries", "stat_failed_moves"):
                raise KeyError(k)
            v = np.asarray(v)
            for eng, (lo, hi) in zip(self.engines, self._bounds):
                eng.write(k, np.ascontiguousarray(v[lo:hi].T) if k == "rng" else v[lo:hi])
        if refresh_obs:
            for eng in self.engines:
                eng.refresh_obs()

    def shelf_xy(self) -> np.ndarray:
        """(B, S, 2) (x, y) of every shelf id, read off the shelf layer (`env.shelfs[j].x/.y`)."""
        grid = self._gather("grid")
        out = np.zeros((self.num_envs, self.n_shelves, 2), np.int32)
        e, ys, xs = np.nonzero(grid[:, 1])
        ids = grid[e, 1, ys, xs]
        out[e, ids - 1, 0] = xs
        out[e, ids - 1, 1] = ys
        return out

    def recalc_grid(self, shelf_xy, refresh_obs: bool = True):
        """Warehouse._recalc_grid (:749-755) from explicit shelf positions + current agent positions."""
        s = np.asarray(shelf_xy, np.int32).reshape(self.num_envs, -1, 2)
        for eng, (lo, hi) in zip(self.engines, self._bounds):
            eng.recalc_grid(s[lo:hi])
            if refresh_obs:
                eng.refresh_obs()

    def refresh_grid(self):
        """`grid` and the five `agent_*` arrays are derived views (rebuilt from the shelf layer and the packed agent records
        the kernels keep): get_state() and a fresh device_tensor(name) are always current; call this to update a grid
        tensor obtained earlier (an agent_* tensor: device_tensor(name) again — same memory, refreshed)."""
        for eng in self.engines:
            eng.refresh_grid()

    def observations(self):
        return self._observations()

    # ------------------------------------------------------------------------------- the rest of Warehouse's public surface
    def seed(self, seed=None):
        """Warehouse.seed (rware/warehouse.py:962-964): re-seed the RNG streams without resetting — env i gets
        `np_random(seed + i)` (the vector convention of reset(seed=...)); None leaves them alone."""
        if seed is None:
            return
        if int(seed) < 0:
            raise ValueError(f"Seed must be a non-negative integer, got {seed!r}")
        lib = getattr(self.engines[0], "lib", None)
        states = np.empty((self.num_envs, 6), np.uint64)
        for i in range(self.num_envs):
            rc = lib.rw_seed_state(int(seed) + i, states[i].ctypes.data)
            assert rc == 0
        self.set_state(refresh_obs=False, rng=states)
        self._seeded = True

    def get_global_image(self, image_layers=(ImageLayer.SHELVES, ImageLayer.GOALS), recompute=False, pad_to_shape=None):
        """Warehouse.get_global_image (:966-1040) for every env: float32 (B, C, H, W) (or (B,) + pad_to_shape), layer by layer
        as the reference builds them — including its cache (`recompute=False` returns the last image, whatever layers it was
        built with) and the transposed AGENT_DIRECTION / AGENT_LOAD layers (IndexError where the reference raises it).
        Host-side, from get_state(): a global-state input for centralised critics, not part of the per-step path."""
        if not recompute and getattr(self, "global_image", None) is not None:
            return self.global_image
        st = self.get_state()
        self.global_image = global_image_from_state(st, self.goals, image_layers, pad_to_shape)
        return self.global_image

    def close(self, **kwargs):
        if self._multi is not None:
            self._multi.close()
            self._multi = None
        for eng in self.engines:
            eng.close()
        self.engines = []
        self._tviews = {}
        self._fast = None          # (the fast path of step() holds the raw engine handle and views of the freed slab)
        self._fast_ok = False
        self._live_actions = None
        if self._pool is not None:
            self._pool.shutdown(wait=False)
            self._pool = None
        self.closed = True

    def close_extras(self, **kwargs):
        self.close()

    # ------------------------------------------------------------------------------- VectorEnv conveniences
    @property
    def unwrapped(self):
        return self

    def get_attr(self, name):
        """Gymnasium VectorEnv.get_attr: the attribute of every sub-env — all envs share one configuration."""
        v = getattr(self, name)
        return tuple(v for _ in range(self.num_envs))

    def call(self, name, *args, **kwargs):
        """Gymnasium VectorEnv.call: method (or attribute) `name` evaluated once, returned per sub-env."""
        v = getattr(self, name)
        r = v(*args, **kwargs) if callable(v) else v
        return tuple(r for _ in range(self.num_envs))

    def set_attr(self, name, values):
        raise NotImplementedError("the batched envs share one compiled configuration; construct a new WarehouseVecEnv instead")

    def render(self):
        """The reference renders with pyglet (rware/rendering.py) — outside the accelerated path.  get_state() has
        everything a renderer needs (grid, agents, queue)."""
        raise NotImplementedError("rendering is outside the accelerated step path; use get_state() with the reference renderer")
