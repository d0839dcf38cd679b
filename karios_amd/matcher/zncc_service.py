"""ZNCC scoring -- drop-in for `karios.matcher.zncc_service.ZNCCService.compute_zncc`.

The reference scores every candidate key point with a pandas `apply` over rows
(`zncc_service.py:162-238`, ~160 us per key point); here the whole frame is scored by one
batched HIP kernel (one wavefront per key point, fp64) with the same rounding, bounds and
NaN rules.
"""
from __future__ import annotations

import logging

import numpy as np
from pandas import DataFrame, Series

from .. import ops

logger = logging.getLogger(__name__)


def _zncc2(img1, img2, u1: int, v1: int, u2: int, v2: int, n: int) -> float:
    """ZNCC between the (2n+1)^2 patches centred at (u1,v1) / (u2,v2) (rows, cols) of two
    images -- same contract as the reference `_zncc2` (zncc_service.py:45-126): ValueError for
    n < 0, IndexError when a window leaves its image, NaN for a zero-variance patch.
    Only n == 21 runs on the GPU kernel (the service's fixed window); it is exposed for tests."""
    if n < 0:
        raise ValueError("Window half-size n must be non-negative")
    img1, img2 = np.asarray(img1), np.asarray(img2)
    h1, w1 = img1.shape
    h2, w2 = img2.shape
    if (u1 - n < 0 or u1 + n >= h1 or v1 - n < 0 or v1 + n >= w1
            or u2 - n < 0 or u2 + n >= h2 or v2 - n < 0 or v2 + n >= w2):
        raise IndexError("Patch window extends beyond image boundaries")
    if n != 21:
        raise NotImplementedError("the HIP ZNCC kernel implements the service window n=21 only")
    # embed the two 43x43 windows in 57x57 chips so that the kernel's bounds rule passes
    chip1 = np.zeros((57, 57), img1.dtype)
    chip2 = np.zeros((57, 57), img2.dtype)
    chip1[7:50, 7:50] = img1[u1 - n:u1 + n + 1, v1 - n:v1 + n + 1]
    chip2[7:50, 7:50] = img2[u2 - n:u2 + n + 1, v2 - n:v2 + n + 1]
    if chip1.dtype != chip2.dtype or chip1.dtype.kind == "f" and chip1.dtype != np.float32:
        chip1, chip2 = chip1.astype(np.float32), chip2.astype(np.float32)
    z = np.zeros(1, np.float32)
    return float(ops.zncc_batch(chip1, chip2, z + 28, z + 28, z, z)[0])


class ZNCCService:
    """Service class to compute ZNCC between reference / monitored patches of each key point."""

    def __init__(self, ctx=None):
        self._chip_size = 57
        self._chip_margin = int((self._chip_size - 1) / 2)
        self._ctx = ctx

    def compute_zncc(self, df: DataFrame, monitored, reference) -> Series:
        """Compute ZNCC for each KP of the given dataframe (reference zncc_service.py:162-184).

        Args:
            df: dataframe with columns x0, y0, dx, dy
            monitored / reference: images exposing `.array` (full resolution, raw dtype),
                `.x_size`, `.y_size`, `.clear_cache()`

        Returns:
            Series with the index of `df`; NaN where the reference skips the point.
        """
        logger.info("Compute ZNCC for %s points", len(df))
        if len(df) == 0:
            score = Series([], index=df.index, dtype=np.float64)
        else:
            cols = [df[c].to_numpy(dtype=np.float32, copy=False) for c in ("x0", "y0", "dx", "dy")]
            values = ops.zncc_batch(reference.array, monitored.array, *cols, ctx=self._ctx)
            score = Series(values, index=df.index, dtype=np.float64)
        monitored.clear_cache()
        reference.clear_cache()
        logger.info("ZNCC computation finish")
        return score

    def compute_mi(self, df: DataFrame, monitored, reference) -> Series:
        """NMI (2*MI/(H(X)+H(Y)), 32 bins, 57x57 chips) for each KP (reference zncc_service.py:240-287)."""
        logger.info("Compute NMI for %s points", len(df))
        if len(df) == 0:
            score = Series([], index=df.index, dtype=np.float64)
        else:
            cols = [df[c].to_numpy(dtype=np.float32, copy=False) for c in ("x0", "y0", "dx", "dy")]
            _, nmi = ops.mi_batch(reference.array, monitored.array, *cols, ctx=self._ctx)
            score = Series(nmi, index=df.index, dtype=np.float64)
        monitored.clear_cache()
        reference.clear_cache()
        logger.info("NMI computation finish")
        return score

    def _extract_chip(self, x: int, y: int, image):
        """57x57 chip around (x, y) (reference zncc_service.py:289-297)."""
        m = self._chip_margin
        return image.array[y - m:y + m + 1, x - m:x + m + 1]
