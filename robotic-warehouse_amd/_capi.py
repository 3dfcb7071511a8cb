"""ctypes binding of include/rware_hip.h (the in-tree equivalent of the stub in INTEGRATION.md).

There is no CPU fallback: if `csrc/librware_hip.so` has not been built this module raises, and
if no HIP device is visible `rw_create` fails with RW_ERR_NO_DEVICE.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIBRARY = os.path.join(_HERE, "csrc", "librware_hip.so")

RW_ABI_VERSION = 3
RW_OK, RW_ERR_INVALID_ARG, RW_ERR_INVALID_ACTION, RW_ERR_HIP, RW_ERR_UNSUPPORTED, RW_ERR_NO_DEVICE, RW_ERR_INDEX, RW_ERR_SELFTEST = 0, -1, -2, -3, -4, -5, -6, -7

BUF = {
    "obs": 0, "rewards": 1, "terminated": 2, "truncated": 3, "grid": 4, "agent_x": 5, "agent_y": 6,
    "agent_dir": 7, "agent_carry": 8, "agent_delivered": 9, "queue": 10, "steps": 11, "inactive": 12,
    "rng": 13, "need_reset": 14, "actions": 15, "features": 16, "agent_msg": 17, "final_obs": 18, "final_features": 19,
    "stat_deliveries": 20, "stat_failed_moves": 21,  # RW_STATS_ON only (empty otherwise)
}
BUF_DTYPE = {
    "obs": np.float32, "rewards": np.float32, "terminated": np.uint8, "truncated": np.uint8,
    "rng": np.uint64, "need_reset": np.uint8, "features": np.float32, "final_obs": np.float32, "final_features": np.float32,
}

RW_STREAM_USE_GIVEN = 1  # rw_stream_flags: `stream` is taken literally, NULL == the device's default stream
RW_OBS_STORES_CACHED, RW_OBS_STORES_STREAM = 2, 4  # rw_stream_flags: keep the observation lines cached / force the non-temporal hint
RW_JIT_OFF, RW_JIT_FORCE = 8, 16  # rw_stream_flags: run-time specialisation (hipRTC) never / always; default: shapes without an exact build, B >= 4096
RW_PIPE_OFF, RW_PIPE_ON = 32, 64  # rw_stream_flags: the chunk-pipelined persistent per-step kernel never / wherever a build exists; default: the engine's measured rule
RW_PRIO_OFF, RW_PRIO_ON = 256, 512  # rw_stream_flags: raised wavefront priority on the chain in front of the first store never / always; default: the engine's measured rule
RW_STATS_ON = 128  # rw_stream_flags: keep the per-env event counters RW_BUF_STAT_DELIVERIES / _FAILED_MOVES (off by default)

AUTORESET = {"disabled": 0, None: 0, "next_step": 1, "same_step": 2}


class RwConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "abi_version", "num_envs", "grid_h", "grid_w", "n_agents", "sensor_range",
        "request_queue_size", "max_inactivity_steps", "max_steps", "reward_type",
        "normalised_coordinates", "autoreset_mode", "n_goals", "device_id",
        "envs_per_workgroup", "threads_per_workgroup", "observation_type", "image_directional",
        "n_image_layers")] + [("image_layers", C.c_int32 * 8), ("msg_bits", C.c_int32), ("stream_flags", C.c_int32),
        ("highways", C.c_void_p), ("goals_xy", C.c_void_p), ("stream", C.c_void_p)]


class RwInfo(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "num_envs", "grid_h", "grid_w", "n_agents", "request_queue_size", "n_shelves", "obs_length",
        "envs_per_workgroup", "threads_per_workgroup", "n_workgroups", "lds_bytes", "device_id",
        "compute_units", "specialised", "wave_priority", "build_kind")] + [
        ("algorithmic_bytes_per_env_step", C.c_int64),
        ("device_name", C.c_char * 128), ("arch_name", C.c_char * 64), ("obs_stores_stream", C.c_int32), ("jit", C.c_int32),
        ("engine_bytes_per_env_step", C.c_int64), ("stagger_ticks", C.c_int32), ("pipe_envs_per_workgroup", C.c_int32),
        ("pipe_workgroups", C.c_int32), ("stats", C.c_int32)]


EXPORTS = (
    "rw_create", "rw_destroy", "rw_last_error", "rw_reset", "rw_step", "rw_step_device",
    "rw_step_many_device", "rw_step_tape_device", "rw_step_tape_device_timed", "rw_refresh_obs", "rw_refresh_grid", "rw_mark_views_stale", "rw_set_stream", "rw_jit_log", "rw_jit_probe", "rw_multi_create", "rw_multi_step_device", "rw_multi_destroy", "rw_sync", "rw_get_buffer", "rw_read", "rw_read_outputs", "rw_write",
    "rw_recalc_grid", "rw_get_info", "rw_seed_state", "rw_event_record", "rw_event_elapsed_ms",
    "rw_abi_version", "rw_debug_timeline", "rw_debug_store_floor", "rw_device_malloc", "rw_device_free", "rw_copy_to_device",
    "rw_copy_to_host", "rw_selftest", "rw_snapshot_create", "rw_snapshot_save", "rw_snapshot_restore", "rw_snapshot_destroy",
)

_libs = {}


def _preload_hip_runtime():
    """Keep ONE HIP runtime in the process.  PyTorch-ROCm wheels bundle their own libamdhip64;
    if that copy and /opt/rocm's are both initialised in one process the second one sees no GPUs
    (and device pointers could not be shared zero-copy).  So when torch is installed, its bundled
    runtime is loaded first and librware_hip.so (NEEDED libamdhip64.so.7) binds to it."""
    import importlib.util

    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return None
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            return C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            return None
    return None


def load(path: str | None = None):
    """dlopen the engine library (default: the in-tree gfx950 build) and declare its prototypes."""
    path = os.path.abspath(path or DEFAULT_LIBRARY)
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} not found: the HIP engine is not built. Run `python -c 'import __graft_entry__ as g; "
            "g.build()'` (or `make -C robotic-warehouse_amd/csrc`). There is no CPU fallback.")
    _preload_hip_runtime()
    lib = C.CDLL(path)
    vp, i32 = C.c_void_p, C.c_int32
    lib.rw_create.argtypes = [C.POINTER(RwConfig), C.POINTER(vp)]
    lib.rw_destroy.argtypes = [vp]
    lib.rw_last_error.argtypes = [vp]
    lib.rw_last_error.restype = C.c_char_p
    lib.rw_reset.argtypes = [vp, vp, vp]
    lib.rw_step.argtypes = [vp, vp]
    lib.rw_step_device.argtypes = [vp, vp]
    lib.rw_step_many_device.argtypes = [vp, vp, i32, vp, vp, vp]
    lib.rw_step_tape_device.argtypes = [vp, vp, i32, i32, i32]
    lib.rw_step_tape_device_timed.argtypes = [vp, vp, i32, i32, i32, i32, i32]
    lib.rw_refresh_obs.argtypes = [vp]
    lib.rw_refresh_grid.argtypes = [vp]
    lib.rw_mark_views_stale.argtypes = [vp]
    lib.rw_set_stream.argtypes = [vp, vp]
    lib.rw_jit_log.argtypes = [vp]
    lib.rw_jit_log.restype = C.c_char_p
    lib.rw_jit_probe.argtypes = [C.POINTER(C.c_int32), C.c_char_p, C.c_char_p, C.c_size_t]
    lib.rw_jit_probe.restype = C.c_int64
    lib.rw_selftest.argtypes = [i32, C.c_char_p, C.c_size_t]
    lib.rw_debug_store_floor.argtypes = [vp, i32, C.POINTER(C.c_float)]
    lib.rw_multi_create.argtypes = [C.POINTER(vp), i32, C.POINTER(vp)]
    lib.rw_multi_step_device.argtypes = [vp, C.POINTER(vp)]
    lib.rw_multi_destroy.argtypes = [vp]
    lib.rw_sync.argtypes = [vp]
    lib.rw_get_buffer.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(C.c_size_t)]
    lib.rw_read.argtypes = [vp, C.c_int, vp, C.c_size_t]
    lib.rw_read_outputs.argtypes = [vp, vp, vp, vp, vp]
    lib.rw_write.argtypes = [vp, C.c_int, vp, C.c_size_t]
    lib.rw_recalc_grid.argtypes = [vp, vp, i32]
    lib.rw_get_info.argtypes = [vp, C.POINTER(RwInfo)]
    lib.rw_seed_state.argtypes = [C.c_uint64, vp]
    lib.rw_event_record.argtypes = [vp, i32]
    lib.rw_event_elapsed_ms.argtypes = [vp, i32, i32, C.POINTER(C.c_float)]
    lib.rw_abi_version.argtypes = []
    lib.rw_debug_timeline.argtypes = [vp, vp, vp, C.POINTER(i32), C.POINTER(i32)]
    lib.rw_device_malloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    lib.rw_device_free.argtypes = [vp, vp]
    lib.rw_copy_to_device.argtypes = [vp, vp, vp, C.c_size_t]
    lib.rw_copy_to_host.argtypes = [vp, vp, vp, C.c_size_t]
    lib.rw_snapshot_create.argtypes = [vp, C.POINTER(vp)]
    lib.rw_snapshot_save.argtypes = [vp, vp]
    lib.rw_snapshot_restore.argtypes = [vp, vp]
    lib.rw_snapshot_destroy.argtypes = [vp, vp]
    for name in EXPORTS:
        if name not in ("rw_last_error", "rw_jit_log", "rw_jit_probe"):
            getattr(lib, name).restype = C.c_int
    if lib.rw_abi_version() != RW_ABI_VERSION:
        raise RuntimeError(f"{path}: ABI version {lib.rw_abi_version()} != {RW_ABI_VERSION}")
    _libs[path] = lib
    return lib


class EngineError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"rware engine error {code}: {msg}")
        self.code = code


class DeviceArray:
    """Borrowed view of an engine buffer; exposes __cuda_array_interface__ so that
    `torch.as_tensor(view, device='cuda')` wraps it without a copy (HIP memory on ROCm torch)."""

    def __init__(self, ptr, shape, dtype, owner):
        self.ptr, self.shape, self.dtype, self._owner = int(ptr), tuple(shape), np.dtype(dtype), owner

    @property
    def __cuda_array_interface__(self):
        return {"shape": self.shape, "typestr": self.dtype.str, "data": (self.ptr, False), "version": 2, "strides": None}


def selftest(device_id: int = 0, library=None):
    """rw_selftest: the on-device check of the toolchain facts the kernels rely on; returns (ok, message)."""
    lib = load(library)
    log = C.create_string_buffer(1024)
    rc = lib.rw_selftest(int(device_id), log, len(log))
    return rc == RW_OK, log.value.decode()


class Engine:
    """One rw_engine: `num_envs` warehouses on one HIP device."""

    def __init__(self, *, num_envs, layout, n_agents, sensor_range, request_queue_size,
                 max_inactivity_steps, max_steps, reward_type, normalised_coordinates=False,
                 autoreset_mode="next_step", device_id=0, envs_per_workgroup=0,
                 threads_per_workgroup=0, stream=None, library=None, observation_type=1,
                 image_layers=(), image_directional=True, msg_bits=0, use_given_stream=False, obs_stores=None, jit=None, pipe=None, stats=False, wave_priority=None):
        self.lib = load(library)
        self._h = C.c_void_p()
        self._arena, self.arena_allocations = {}, 0  # rollout_host's device tapes (grow-only; freed in close())
        hw = np.ascontiguousarray(layout.highways, dtype=np.uint8)
        goals = np.ascontiguousarray(np.asarray(layout.goals, dtype=np.int32).reshape(-1))
        cfg = RwConfig(
            RW_ABI_VERSION, int(num_envs), int(layout.grid_size[0]), int(layout.grid_size[1]), int(n_agents),
            int(sensor_range), int(request_queue_size), int(max_inactivity_steps or 0), int(max_steps or 0),
            int(reward_type), int(bool(normalised_coordinates)), AUTORESET[autoreset_mode], len(layout.goals),
            int(device_id), int(envs_per_workgroup), int(threads_per_workgroup),
            int(observation_type), int(bool(image_directional)), len(image_layers),
            (C.c_int32 * 8)(*[int(l) for l in image_layers]), int(msg_bits),
            (RW_STREAM_USE_GIVEN if use_given_stream else 0) | {None: 0, "auto": 0, "cached": RW_OBS_STORES_CACHED, "stream": RW_OBS_STORES_STREAM}[obs_stores]
            | {None: 0, "auto": 0, False: RW_JIT_OFF, "off": RW_JIT_OFF, True: RW_JIT_FORCE, "force": RW_JIT_FORCE}[jit]
            | {None: 0, "auto": 0, False: RW_PIPE_OFF, "off": RW_PIPE_OFF, True: RW_PIPE_ON, "on": RW_PIPE_ON}[pipe]
            | (RW_STATS_ON if stats else 0)
            | {None: 0, "auto": 0, False: RW_PRIO_OFF, "off": RW_PRIO_OFF, True: RW_PRIO_ON, "on": RW_PRIO_ON}[wave_priority],
            hw.ctypes.data, goals.ctypes.data, C.c_void_p(stream or 0))
        rc = self.lib.rw_create(C.byref(cfg), C.byref(self._h))
        if rc != RW_OK:
            raise EngineError(rc, (self.lib.rw_last_error(None) or b"").decode())
        self.info = RwInfo()
        self._check(self.lib.rw_get_info(self._h, C.byref(self.info)))
        i = self.info
        self.B, self.N, self.Q, self.L, self.S = i.num_envs, i.n_agents, i.request_queue_size, i.obs_length, i.n_shelves
        self.H, self.W = i.grid_h, i.grid_w
        self.M = int(msg_bits)
        win = 2 * int(sensor_range) + 1
        obs_shape = (self.B, self.N, self.L) if int(observation_type) == 1 else (self.B, self.N, self.L // (win * win), win, win)
        self.shapes = {
            "features": (self.B, self.N, 6), "final_features": (self.B, self.N, 6),
            "obs": obs_shape, "final_obs": obs_shape, "rewards": (self.B, self.N), "terminated": (self.B,),
            "truncated": (self.B,), "grid": (self.B, 2, self.H, self.W), "agent_x": (self.B, self.N),
            "agent_y": (self.B, self.N), "agent_dir": (self.B, self.N), "agent_carry": (self.B, self.N),
            "agent_delivered": (self.B, self.N), "queue": (self.B, self.Q), "steps": (self.B,),
            "inactive": (self.B,), "rng": (6, self.B), "need_reset": (self.B,),
            "actions": (self.B, self.N, 1 + self.M) if self.M else (self.B, self.N), "agent_msg": (self.B, self.N),
            "stat_deliveries": (self.B,), "stat_failed_moves": (self.B,),
        }
        self.stats = bool(i.stats)

    def _check(self, rc):
        if rc != RW_OK:
            msg = (self.lib.rw_last_error(self._h) or b"").decode()
            if rc == RW_ERR_INVALID_ACTION:
                raise ValueError(msg or "invalid action")
            if rc == RW_ERR_INDEX:  # the reference's IndexError in _make_img_obs (rware/warehouse.py:552,558)
                raise IndexError(msg or "image layer index out of bounds")
            raise EngineError(rc, msg)

    def close(self):
        if self._h:
            for p, _ in self._arena.values():
                self.lib.rw_device_free(self._h, p)
            self._arena = {}
            self.lib.rw_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # hot path -----------------------------------------------------------------------------
    def reset(self, seeds=None, mask=None):
        s = None if seeds is None else np.ascontiguousarray(np.asarray(seeds, dtype=np.uint64).reshape(self.B))
        m = None if mask is None else np.ascontiguousarray(np.asarray(mask).astype(np.uint8).reshape(self.B))
        self._check(self.lib.rw_reset(self._h, None if s is None else s.ctypes.data, None if m is None else m.ctypes.data))

    def step_host(self, actions_i32):
        a = np.ascontiguousarray(actions_i32, dtype=np.int32)
        assert a.size == self.B * self.N * (1 + self.M)
        self._check(self.lib.rw_step(self._h, a.ctypes.data))

    def step_device(self, dev_ptr):
        self._check(self.lib.rw_step_device(self._h, C.c_void_p(int(dev_ptr))))

    def step_tape_device(self, tape_ptr, tape_steps, first, n_steps):
        """n_steps per-step launches (rw_step_device each) from a device action tape, issued by one native loop."""
        self._check(self.lib.rw_step_tape_device(self._h, C.c_void_p(int(tape_ptr)), int(tape_steps), int(first), int(n_steps)))

    def step_tape_device_timed(self, tape_ptr, tape_steps, first, n_steps, start_slot, stop_slot):
        """step_tape_device with the timing events attached to the first / last launch (no marker packets)."""
        self._check(self.lib.rw_step_tape_device_timed(self._h, C.c_void_p(int(tape_ptr)), int(tape_steps), int(first),
                                                       int(n_steps), int(start_slot), int(stop_slot)))

    def step_many_device(self, dev_ptr, n_steps, obs_tape=0, reward_tape=0, terminated_tape=0):
        self._check(self.lib.rw_step_many_device(self._h, C.c_void_p(int(dev_ptr)), int(n_steps),
                                                 C.c_void_p(int(obs_tape)), C.c_void_p(int(reward_tape)),
                                                 C.c_void_p(int(terminated_tape))))

    def rollout_host(self, actions, want_obs=True):
        """`T` fused steps (one launch) from a host action tape (T, B, N); returns host tapes
        (obs (T,B,N,L) or None, rewards (T,B,N), terminated (T,B))."""
        a = np.ascontiguousarray(actions, dtype=np.int32).reshape(-1, self.B, self.N * (1 + self.M))
        T = a.shape[0]
        obs = np.empty((T,) + self.shapes["obs"], np.float32) if want_obs else None
        rew = np.empty((T, self.B, self.N), np.float32)
        term = np.empty((T, self.B), np.uint8)
        # device tapes from the engine's arena: grow-only buffers kept for the engine's lifetime, so a training loop that
        # calls rollout() with the same T does no device allocation after the first call
        d_a = self._arena_buf("actions", a.nbytes)
        self._check(self.lib.rw_copy_to_device(self._h, d_a, a.ctypes.data, a.nbytes))
        d_o = self._arena_buf("obs", obs.nbytes) if want_obs else C.c_void_p(0)
        d_r, d_t = self._arena_buf("rewards", rew.nbytes), self._arena_buf("terminated", term.nbytes)
        self._check(self.lib.rw_step_many_device(self._h, d_a, T, d_o, d_r, d_t))
        if want_obs:
            self._check(self.lib.rw_copy_to_host(self._h, obs.ctypes.data, d_o, obs.nbytes))
        self._check(self.lib.rw_copy_to_host(self._h, rew.ctypes.data, d_r, rew.nbytes))
        self._check(self.lib.rw_copy_to_host(self._h, term.ctypes.data, d_t, term.nbytes))
        return obs, rew, term

    def _arena_buf(self, name, nbytes):
        ent = self._arena.get(name)
        if ent is None or ent[1] < nbytes:
            if ent is not None:
                self._check(self.lib.rw_device_free(self._h, ent[0]))
                del self._arena[name]
            p = C.c_void_p()
            self._check(self.lib.rw_device_malloc(self._h, nbytes, C.byref(p)))
            ent = self._arena[name] = (p, nbytes)
            self.arena_allocations += 1
        return ent[0]

    def snapshot(self, into=None):
        """Device-resident copy of the whole env state (see rw_snapshot_*); returns an opaque handle."""
        h = into
        if h is None:
            h = C.c_void_p()
            self._check(self.lib.rw_snapshot_create(self._h, C.byref(h)))
        self._check(self.lib.rw_snapshot_save(self._h, h))
        return h

    def restore(self, handle):
        self._check(self.lib.rw_snapshot_restore(self._h, handle))

    def free_snapshot(self, handle):
        self._check(self.lib.rw_snapshot_destroy(self._h, handle))

    def refresh_obs(self):
        self._check(self.lib.rw_refresh_obs(self._h))

    def refresh_grid(self):
        """Brings the exported int32 grid (a derived view) up to date with the steps enqueued so far."""
        self._check(self.lib.rw_refresh_grid(self._h))

    def jit_log(self) -> str:
        return (self.lib.rw_jit_log(self._h) or b"").decode()

    def set_stream(self, stream_handle):
        """Later launches go to `stream_handle` (a hipStream_t as int; 0 = the default stream)."""
        self._check(self.lib.rw_set_stream(self._h, C.c_void_p(int(stream_handle or 0))))

    def mark_views_stale(self):
        """Steps ran that the host side did not see (a replayed HIP graph): the derived views are rebuilt when next asked for."""
        self._check(self.lib.rw_mark_views_stale(self._h))

    def release_arena(self):
        """Frees the device tapes rollout_host() keeps between calls (grow-only otherwise, for the engine's lifetime)."""
        for p, _ in self._arena.values():
            self._check(self.lib.rw_device_free(self._h, p))
        self._arena = {}

    def sync(self):
        self._check(self.lib.rw_sync(self._h))

    # buffers ------------------------------------------------------------------------------
    def read(self, name) -> np.ndarray:
        out = np.empty(self.shapes[name], dtype=BUF_DTYPE.get(name, np.int32))
        self._check(self.lib.rw_read(self._h, BUF[name], out.ctypes.data, out.nbytes))
        return out

    def read_outputs(self, want_features=False):
        """obs, rewards, terminated (uint8), features-or-None as fresh host arrays: one C call, one synchronisation."""
        obs = np.empty(self.shapes["obs"], np.float32)
        rew = np.empty(self.shapes["rewards"], np.float32)
        term = np.empty(self.shapes["terminated"], np.uint8)
        feat = np.empty(self.shapes["features"], np.float32) if want_features else None
        self._check(self.lib.rw_read_outputs(self._h, obs.ctypes.data, rew.ctypes.data, term.ctypes.data,
                                             feat.ctypes.data if want_features else None))
        return obs, rew, term, feat

    def write(self, name, array):
        a = np.ascontiguousarray(array, dtype=BUF_DTYPE.get(name, np.int32)).reshape(self.shapes[name])
        self._check(self.lib.rw_write(self._h, BUF[name], a.ctypes.data, a.nbytes))

    def device_array(self, name) -> DeviceArray:
        ptr, nbytes = C.c_void_p(), C.c_size_t()
        self._check(self.lib.rw_get_buffer(self._h, BUF[name], C.byref(ptr), C.byref(nbytes)))
        return DeviceArray(ptr.value or 0, self.shapes[name], BUF_DTYPE.get(name, np.int32), self)

    def recalc_grid(self, shelf_xy):
        s = np.ascontiguousarray(shelf_xy, dtype=np.int32).reshape(self.B, -1, 2)
        self._check(self.lib.rw_recalc_grid(self._h, s.ctypes.data, s.shape[1]))

    def debug_timeline(self, actions_dev_ptr) -> np.ndarray:
        """uint64 [n_workgroups, n_marks] phase stamps (10 ns ticks) of one step; profiling aid."""
        nwg, nm = C.c_int32(), C.c_int32()
        self._check(self.lib.rw_debug_timeline(self._h, C.c_void_p(int(actions_dev_ptr)), None, C.byref(nwg), C.byref(nm)))
        out = np.zeros((nwg.value, nm.value), dtype=np.uint64)
        self._check(self.lib.rw_debug_timeline(self._h, C.c_void_p(int(actions_dev_ptr)), out.ctypes.data, None, None))
        return out

    def debug_store_floor(self, n_launches=2000) -> float:
        """ms per launch of a kernel that only writes one step's observations (same geometry and store instruction): measurement aid."""
        ms = C.c_float()
        self._check(self.lib.rw_debug_store_floor(self._h, int(n_launches), C.byref(ms)))
        return float(ms.value)

    def event_record(self, slot):
        self._check(self.lib.rw_event_record(self._h, slot))

    def event_elapsed_ms(self, a, b) -> float:
        ms = C.c_float()
        self._check(self.lib.rw_event_elapsed_ms(self._h, a, b, C.byref(ms)))
        return float(ms.value)


class MultiEngine:
    """rw_multi: one C call per step for all the engines of a single-process multi-device env (launcher thread per engine)."""

    def __init__(self, engines):
        self.engines, self.lib = list(engines), engines[0].lib
        n = len(self.engines)
        hs = (C.c_void_p * n)(*[e._h for e in self.engines])
        self._h = C.c_void_p()
        rc = self.lib.rw_multi_create(hs, n, C.byref(self._h))
        if rc != RW_OK:
            raise EngineError(rc, "rw_multi_create failed")
        self._ptrs = (C.c_void_p * n)()

    def step_device(self, dev_ptrs):
        p = self._ptrs
        for k, v in enumerate(dev_ptrs):
            p[k] = v
        rc = self.lib.rw_multi_step_device(self._h, p)
        if rc != RW_OK:
            # (engine 0's message names the engine that failed this round — rw_multi_step_device writes it there on every path;
            #  should it be empty all the same, the first engine that has one)
            msg = (self.lib.rw_last_error(self.engines[0]._h) or b"").decode()
            if not msg:
                msg = next((m for m in ((self.lib.rw_last_error(e._h) or b"").decode() for e in self.engines[1:]) if m), "")
            raise EngineError(rc, msg or "rw_multi_step_device failed")

    def close(self):
        if self._h:
            self.lib.rw_multi_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def seed_state(seed: int, library=None) -> np.ndarray:
    out = np.zeros(6, dtype=np.uint64)
    rc = load(library).rw_seed_state(C.c_uint64(int(seed)), out.ctypes.data)
    assert rc == RW_OK
    return out


def jit_probe(*, sensor_range, H, W, N, Q, S, E, msg_bits=0, obs=0, layers=(), directional=True, nt=1, arch="gfx950", library=None):
    """Compile (or find cached) the run-time specialised build of a shape without a device; returns (code bytes or -1, log)."""
    packed = 0
    for k, l in enumerate(layers):
        packed |= int(l) << (4 * k)
    shape = (C.c_int32 * 15)(sensor_range, H, W, N, Q, S, E, 256, msg_bits, 1 if S > 255 else 0, obs, len(layers), packed,
                             (1 if directional else 0) if obs == 1 else -1, nt)
    log = C.create_string_buffer(4096)
    n = load(library).rw_jit_probe(shape, arch.encode(), log, 4096)
    return int(n), log.value.decode(errors="replace")
