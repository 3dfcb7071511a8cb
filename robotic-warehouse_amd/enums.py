"""Enums of the RWARE step path, value-compatible with rware/warehouse.py:31-56."""
from enum import Enum


class Action(Enum):  # rware/warehouse.py:31-36
    NOOP = 0
    FORWARD = 1
    LEFT = 2
    RIGHT = 3
    TOGGLE_LOAD = 4


class Direction(Enum):  # rware/warehouse.py:39-43
    UP = 0
    DOWN = 1
    LEFT = 2
    RIGHT = 3


class RewardType(Enum):  # rware/warehouse.py:46-49
    GLOBAL = 0
    INDIVIDUAL = 1
    TWO_STAGE = 2


class ObservationType(Enum):  # rware/warehouse.py:52-56
    DICT = 0
    FLATTENED = 1
    IMAGE = 2
    IMAGE_DICT = 3


def enum_value(v):
    """Accepts this package's enums, the reference's enums (same values) or plain ints."""
    return int(getattr(v, "value", v))
