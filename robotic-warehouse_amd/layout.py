"""Host-side warehouse layout, built once per config and handed to the engine as plain arrays.

Mirrors Warehouse._make_layout_from_params (rware/warehouse.py:294-326) and
_make_layout_from_str (:328-350): grid size, `highways` mask, goal cells; shelves spawn on
every non-highway cell (:771-778) and the observation length follows :432-443.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np


@dataclass(frozen=True)
class Layout:
    grid_size: tuple          # (H, W)
    highways: np.ndarray      # uint8 [H, W]
    goals: tuple              # ((x, y), ...) in reward order

    @property
    def n_shelves(self) -> int:
        return int((self.highways == 0).sum())


def layout_from_params(shelf_columns: int, shelf_rows: int, column_height: int) -> Layout:
    assert shelf_columns % 2 == 1, "Only odd number of shelf columns is supported"
    h = (column_height + 1) * shelf_rows + 2
    w = (2 + 1) * shelf_columns + 1
    ys, xs = np.mgrid[0:h, 0:w]
    mid = w // 2
    highways = (
        (xs % 3 == 0)
        | (ys % (column_height + 1) == 0)
        | (ys == h - 1)
        | ((ys > h - (column_height + 3)) & ((xs == mid - 1) | (xs == mid)))
    ).astype(np.uint8)
    return Layout((h, w), np.ascontiguousarray(highways), ((mid - 1, h - 1), (mid, h - 1)))


def layout_from_str(layout: str) -> Layout:
    rows = layout.strip().replace(" ", "").split("\n")
    w = len(rows[0])
    for r in rows:
        assert len(r) == w, "Layout must be rectangular"
    highways = np.zeros((len(rows), w), dtype=np.uint8)
    goals = []
    for y, r in enumerate(rows):
        for x, ch in enumerate(r):
            ch = ch.lower()
            assert ch in "gx.", f"unknown layout character {ch!r}"
            if ch == "g":
                goals.append((x, y))
            highways[y, x] = ch != "x"
    assert len(goals) >= 1, "At least one goal is required"
    return Layout((len(rows), w), highways, tuple(goals))


def obs_length(sensor_range: int, msg_bits: int = 0) -> int:
    cells = (1 + 2 * sensor_range) ** 2
    return (4 + 4) + cells * (1 + 4 + msg_bits) + cells * 2
