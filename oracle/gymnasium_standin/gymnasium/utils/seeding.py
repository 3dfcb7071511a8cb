"""TEST-ONLY stand-in (see ../__init__.py) for gymnasium.utils.seeding."""
import numpy as np


def np_random(seed=None):
    """Gymnasium's published definition: SeedSequence(seed) -> PCG64 -> Generator."""
    if seed is not None and not (isinstance(seed, (int, np.integer)) and seed >= 0):
        raise ValueError(f"Seed must be a non-negative integer, got {seed!r}")
    seed_seq = np.random.SeedSequence(seed)
    np_seed = seed_seq.entropy
    rng = np.random.Generator(np.random.PCG64(seed_seq))
    return rng, np_seed
