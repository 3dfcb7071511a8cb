"""The oracle's numpy-RNG restatement (PCG64 + buffered 32-bit half, Lemire bounded draw,
Floyd + Fisher-Yates `choice(replace=False)`, SeedSequence seeding) checked against numpy
itself (the third-party dependency the reference calls at rware/warehouse.py:781-800,916)."""
import ctypes as C

import numpy as np
import pytest

from rware_oracle import lib, seed_state


def np_state(gen):
    st = gen.bit_generator.state
    s, inc = st["state"]["state"], st["state"]["inc"]
    m = (1 << 64) - 1
    return np.array([s >> 64, s & m, inc >> 64, inc & m, st["has_uint32"], st["uinteger"]], dtype=np.uint64)


@pytest.mark.parametrize("seed", [0, 1, 2, 12345, 2**31 - 1, 2**32 - 1, 2**32, 2**40 + 17, 2**63 + 5])
def test_seedsequence_pcg64_state(seed):
    want = np_state(np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed))))
    assert np.array_equal(seed_state(seed), want)


def test_bounded_matches_integers_and_choice_scalar():
    for seed in range(40):
        gen = np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed)))
        st = seed_state(seed)
        rs = np.random.default_rng(seed)
        for _ in range(200):
            n = int(rs.integers(1, 500))
            if rs.random() < 0.5:
                want = int(gen.integers(0, n))
            else:
                want = int(gen.choice(np.arange(n)))
            got = lib().orc_rng_bounded(st.ctypes.data, n - 1)
            assert got == want
        assert np.array_equal(st, np_state(gen))


def test_choice_without_replacement_matches_numpy():
    for seed in range(60):
        gen = np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed)))
        st = seed_state(seed)
        rs = np.random.default_rng(1000 + seed)
        for _ in range(50):
            pop = int(rs.integers(1, 600))
            k = int(rs.integers(0, min(pop, 20) + 1))
            want = gen.choice(np.arange(pop), size=k, replace=False)
            out = np.zeros(max(k, 1), np.int32)
            lib().orc_rng_choice(st.ctypes.data, pop, k, out.ctypes.data)
            assert np.array_equal(out[:k], want), (seed, pop, k)
            # interleave a sized with-replacement draw (directions: choice(list, size=N))
            n = int(rs.integers(1, 8))
            want2 = gen.choice(4, size=n)
            got2 = [lib().orc_rng_bounded(st.ctypes.data, 3) for _ in range(n)]
            assert list(want2) == got2
        assert np.array_equal(st, np_state(gen))
