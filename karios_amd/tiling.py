"""Tile geometry and Laplacian kernel-size specifications shared by the matcher, the resident pipeline and the
multi-GPU scheduler.

The reference walks its images in square boxes, x outer / y inner, clipping the last box of a row or column and skipping the
columns left of `xStart` (`karios/matcher/klt.py:220-249`); the boxes are independent work units (no halo, per-box
stretch / threshold / maxCorners), which is what `karios_amd.parallel` shards over GPUs.
"""
from __future__ import annotations

from typing import NamedTuple

AUTO_KSIZE_CANDIDATES = (3, 5, 7, 9, 11)   # reference klt.py:39


class Tile(NamedTuple):
    """One box of the image pair (pixels): unpacks as (x_off, y_off, x_size, y_size)."""
    x_off: int
    y_off: int
    x_size: int
    y_size: int


def tile_grid(x_size: int, y_size: int, tile_size: int, x_start: int = 0) -> list[Tile]:
    """Boxes in the order `KLT.match` visits them."""
    if tile_size <= 0:
        raise ValueError(f"tile_size must be positive, got {tile_size}")
    columns = [x for x in range(0, x_size, tile_size) if x >= x_start]
    rows = range(0, y_size, tile_size)
    return [Tile(x, y, min(tile_size, x_size - x), min(tile_size, y_size - y)) for x in columns for y in rows]


def kernel_sizes(spec) -> tuple[int, int]:
    """`laplacian_kernel_size` as configured (an int, or a mapping with "mon" and / or "ref") -> (mon, ref).
    A mapping that names only one image applies that size to both; an empty mapping means 1 (klt.py:431-432)."""
    if not isinstance(spec, dict):
        return spec, spec
    mon, ref = spec.get("mon"), spec.get("ref")
    if mon is None and ref is None:
        return 1, 1
    return (ref if mon is None else mon), (mon if ref is None else ref)
