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


class ImageLayer(Enum):  # rware/warehouse.py:59-70
    SHELVES = 0
    REQUESTS = 1
    AGENTS = 2
    AGENT_DIRECTION = 3  # the reference writes it with transposed indices, layer[ag.x, ag.y] (:552): reproduced as is
    AGENT_LOAD = 4       # same (:558)
    GOALS = 5
    ACCESSIBLE = 6


DEFAULT_IMAGE_LAYERS = (ImageLayer.SHELVES, ImageLayer.REQUESTS, ImageLayer.AGENTS, ImageLayer.GOALS,
                        ImageLayer.ACCESSIBLE)  # rware/warehouse.py:160-166


def enum_value(v):
    """Accepts this package's enums, the reference's enums (same values) or plain ints."""
    return int(getattr(v, "value", v))
