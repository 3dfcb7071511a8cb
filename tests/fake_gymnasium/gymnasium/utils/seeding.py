"""TEST INFRASTRUCTURE — gymnasium.utils.seeding.np_random as Gymnasium >= 0.26 defines it:
SeedSequence(seed) -> PCG64 -> Generator; returns (generator, the seed actually used)."""
import numpy as np


def np_random(seed=None):
    if seed is not None and not (isinstance(seed, int) and seed >= 0):
        if isinstance(seed, int) is False:
            raise ValueError(f"Seed must be a python integer, actual type: {type(seed)}")
        raise ValueError(f"Seed must be greater or equal to zero, actual value: {seed}")
    seed_seq = np.random.SeedSequence(seed)
    np_seed = seed_seq.entropy
    return np.random.Generator(np.random.PCG64(seed_seq)), np_seed
