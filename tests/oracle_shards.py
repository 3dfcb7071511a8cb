"""TEST INFRASTRUCTURE: the C oracle over K shards of a batch on K host threads (the oracle is single-threaded C called through
ctypes, which drops the GIL), for the parity runs at the largest single-GPU configurations — 65536 / 131072 envs, 1.5 GB of
observations per step — where one thread would take seconds per step.  Envs are independent (rware/warehouse.py:804-946 touches
only `self`), shard k seeds its envs SeedSequence(seed + lo_k + i): the same streams as one oracle over the whole batch."""
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from rware_oracle import OracleVecEnv


def checksum_weights(n):
    """Distinct small integer weights per observation slot: the per-env dot product obs . w is an integer below 2^53 (observations
    are 0 / 1 and coordinates below 32), hence EXACT in float64 whatever the order of summation — on the GPU (torch) and in numpy."""
    return ((np.arange(n, dtype=np.int64) * 2654435761) % 65521 + 1).astype(np.float64)


class ShardedOracle:
    def __init__(self, B, shards, **kw):
        assert B % shards == 0
        self.B, self.per = B, B // shards
        self.parts = [OracleVecEnv(self.per, **kw) for _ in range(shards)]
        self.pool = ThreadPoolExecutor(shards)
        self.N, self.L = self.parts[0].N, self.parts[0].L
        self.w = checksum_weights(self.N * self.L)

    def _cs(self, obs):
        return obs.reshape(obs.shape[0], -1).astype(np.float64) @ self.w

    def reset(self, seed):
        """-> per-env checksums of the first observations"""
        return np.concatenate(list(self.pool.map(lambda kp: self._cs(kp[1].reset(seed=seed + kp[0] * self.per)), enumerate(self.parts))))

    def step(self, actions, mode="next_step"):
        """-> (per-env observation checksums, rewards, done) of one vector-env step of the whole batch"""
        def one(kp):
            k, p = kp
            o, r, d = p.step_autoreset(actions[k * self.per:(k + 1) * self.per], mode)
            return self._cs(o), r, d
        out = list(self.pool.map(one, enumerate(self.parts)))
        return tuple(np.concatenate([o[j] for o in out]) for j in range(3))

    def get_state(self):
        st = [p.get_state() for p in self.parts]
        return {k: np.concatenate([s[k] for s in st]) for k in st[0]}

    def close(self):
        self.pool.shutdown()
