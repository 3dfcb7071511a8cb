"""Pins the CPU oracle (oracle/rware_oracle.c) against the golden vectors the unmodified
reference produced (tests/golden/generate_golden.py): every state field, PCG64 state,
FLATTENED obs, rewards and done, bit-for-bit, across autoresets."""
import pytest

import golden_util as gu
from rware_oracle import OracleVecEnv


@pytest.mark.parametrize("name", gu.fixture_names())
def test_oracle_matches_reference_golden(name):
    meta, z = gu.load_fixture(name)
    env = OracleVecEnv(meta["E"], **gu.ctor_kwargs(meta))
    assert gu.replay(env, meta, z) == meta["T"]


def test_fixtures_present():
    assert len(gu.fixture_names()) >= 10
