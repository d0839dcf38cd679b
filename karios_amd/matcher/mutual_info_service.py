"""Mutual-information scoring -- drop-in for `karios.matcher.mutual_info_service.MutualInfoService`
(`compute_mutual_info`, Studholme NMI in [1, 2]).  The reference scores each key point with a pandas `apply`
(~0.7 ms per key point); here one HIP kernel builds the 32x32 joint histogram of every key point's 57x57
chips in LDS.  `ZNCCService.compute_mi` (the [0, 1] variant) shares the kernel.
"""
from __future__ import annotations

import logging

import numpy as np
from pandas import DataFrame, Series

from .. import ops

logger = logging.getLogger(__name__)


class MutualInfoService:
    """Service class to compute normalized mutual information between two image patches."""

    def __init__(self, ctx=None):
        self._chip_size = 57
        self._chip_margin = int((self._chip_size - 1) / 2)
        self._ctx = ctx

    def compute_mutual_info(self, df: DataFrame, monitored, reference) -> Series:
        """Normalized mutual information for each KP of the dataframe (reference mutual_info_service.py:73-97).

        Returns:
            Series with the index of `df`; NaN where the reference skips the point."""
        logger.info("Compute mutual information for %s points", len(df))
        if len(df) == 0:
            score = Series([], index=df.index, dtype=np.float64)
        else:
            cols = [df[c].to_numpy(dtype=np.float32, copy=False) for c in ("x0", "y0", "dx", "dy")]
            st, _ = ops.mi_batch(reference.array, monitored.array, *cols, ctx=self._ctx)
            score = Series(st, index=df.index, dtype=np.float64)
        monitored.clear_cache()
        reference.clear_cache()
        logger.info("Mutual information computation finish")
        return score
