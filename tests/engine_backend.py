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
    subprocess.check_call(["make", "-s", "-j8", "-C", EMU_DIR], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return EMU_LIB


class EngineBackend:
    def __init__(self, num_envs, library=None, autoreset_mode="next_step", **kwargs):
        self.env = rware_amd.WarehouseVecEnv(num_envs, autoreset_mode=autoreset_mode, library=library, **kwargs)
        self.mode = autoreset_mode

    @staticmethod
    def _obs(o):  # IMAGE_DICT -> (image, features), the form the replay harness compares
        return (o["image"], o["features"]) if isinstance(o, dict) else o

    def reset(self, seed=None, mask=None):
        obs, _ = self.env.reset(seed=seed, mask=mask)
        return self._obs(obs)

    def step_autoreset(self, actions, mode):
        assert mode == self.mode
        obs, rew, term, trunc, _ = self.env.step(actions)
        assert not trunc.any()
        return self._obs(obs), rew, term

    def get_state(self):
        return self.env.get_state()
