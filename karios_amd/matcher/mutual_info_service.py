"""Studholme normalised mutual information of the matched key points on MI355X, behind
`karios.matcher.mutual_info_service.MutualInfoService` (reference `mutual_info_service.py:32-130`): per key point the 57x57
chips around `(int(x0), int(y0))` / `(round(x0 + dx), round(y0 + dy))`, a 32x32 joint histogram, (H(X) + H(Y)) / H(X, Y) in
[1, 2]; NaN where a chip leaves its image or the joint entropy vanishes.  The reference spends ~0.7 ms per key point in a
pandas `apply`; here one kernel launch builds every key point's histogram in LDS (k_mi.hip), shared with
`ZNCCService.compute_mi`.
"""
from __future__ import annotations

import logging

import numpy as np
from pandas import DataFrame, Series

from .zncc_service import CHIP_SIZE, _common_pixel_type, _kernel_ready, _mi_scores

logger = logging.getLogger(__name__)


class MutualInfoService:
    """Scores the key points of a frame against the raw (full-resolution) images."""

    def __init__(self, ctx=None):
        self._chip_size = CHIP_SIZE
        self._chip_margin = (CHIP_SIZE - 1) // 2
        self._ctx = ctx

    def compute_mutual_info(self, df: DataFrame, monitored, reference) -> Series:
        """Score per row of `df` (columns x0, y0, dx, dy) on the frame's index; NaN where the reference skips the row."""
        try:
            values = np.empty(0, np.float64)
            if len(df):
                ref, mon = np.asarray(reference.array), np.asarray(monitored.array)
                if not _kernel_ready(df, ref, mon):
                    ref, mon = _common_pixel_type(ref, mon)
                values = _mi_scores(df, ref, mon, self._ctx, rasters=(monitored, reference))[0]
        finally:
            from ..resident import keep_shared_across
            with keep_shared_across(monitored, reference):      # (the service's own closing call: nothing was edited since the upload)
                monitored.clear_cache()
                reference.clear_cache()
        logger.info("mutual information: %d key points scored", len(df))
        return Series(values, index=df.index, dtype=np.float64)
