"""Adapters that let the shared replay harness (golden_util.replay) drive the product's
WarehouseVecEnv — either through the real gfx950 library (GPU tests) or through the
host-thread emulation build of the same sources (tests/emu, CPU-only container)."""
import os
import subprocess

import numpy as np

import rware_amd

EMU_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu")
EMU_LIB = os.path.join(EMU_DIR, "librware_emu.so")


def build_emu() -> str:
    # RWARE_EMU_LIB: a prebuilt emulation library to use instead (oracle/sanitize.sh points the emulated tests at its ASAN build)
    other = os.environ.get("RWARE_EMU_LIB")
    if other:
        return other
    subprocess.check_call(["make", "-s", "-j8", "-C", EMU_DIR], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return EMU_LIB


class EngineBackend:
    """`tile` > 1 runs `tile` identical copies of the `num_envs` envs side by side (same seeds, same actions) and hands
    back the first copy after checking that all copies agree: the exact-shape kernel builds need batches that are a
    multiple of their workgroup size (8 or 16 envs), the golden fixtures hold 2-4 envs."""

    def __init__(self, num_envs, library=None, autoreset_mode="next_step", tile=1, **kwargs):
        self.E, self.tile = num_envs, tile
        self.env = rware_amd.WarehouseVecEnv(num_envs * tile, autoreset_mode=autoreset_mode, library=library, **kwargs)
        self.mode = autoreset_mode

    def _first(self, a):
        a = np.asarray(a)
        if self.tile > 1:
            t = a.reshape((self.tile, self.E) + a.shape[1:])
            assert all(np.array_equal(t[0], t[k]) for k in range(1, self.tile)), "the tiled copies disagree"
            a = t[0]
        return a

    def _obs(self, o):  # IMAGE_DICT -> (image, features), the form the replay harness compares
        return (self._first(o["image"]), self._first(o["features"])) if isinstance(o, dict) else self._first(o)

    def reset(self, seed=None, mask=None):
        if self.tile > 1:
            assert np.isscalar(seed) and mask is None
            seed = np.tile(np.uint64(seed) + np.arange(self.E, dtype=np.uint64), self.tile)
        obs, _ = self.env.reset(seed=seed, mask=mask)
        return self._obs(obs)

    def step_autoreset(self, actions, mode):
        assert mode == self.mode
        a = np.asarray(actions)
        if self.tile > 1:
            a = np.tile(a, (self.tile,) + (1,) * (a.ndim - 1))
        obs, rew, term, trunc, _ = self.env.step(a)
        assert not trunc.any()
        return self._obs(obs), self._first(rew), self._first(term)

    def get_state(self):
        return {k: self._first(v) for k, v in self.env.get_state().items()}
