"""Large-offset pre-aligner -- drop-in for `karios.matcher.large_offset.LargeOffsetMatcher`
plus the thresholding / shifting step of `KariosAPI._detect_large_offset`
(`karios/api/core.py:739-786`).
"""
from __future__ import annotations

import logging

import numpy as np

from .. import ops

logger = logging.getLogger(__name__)


class LargeOffsetMatcher:
    """Class to compute row/col offset between 2 images (reference large_offset.py:25-41)."""

    def __init__(self, reference_image, monitored_image, ctx=None):
        self._ref = reference_image
        self._mon = monitored_image
        self._ctx = ctx

    def match(self):
        """Computes row/col offset between 2 images.

        Returns:
            array([row, col]) float64 holding integers (y/x offset)."""
        # ref and mon parameters are deliberately inverted (reference large_offset.py:38)
        return ops.phase_cross_correlation(self._mon.array, self._ref.array, ctx=self._ctx)


def detect_large_offset(reference_image, monitored_image, bias_correction_min_threshold=2, ctx=None,
                        emulate_gdt_byte_write=False):
    """`_detect_large_offset` numeric part (core.py:751-776): phase correlation, per-axis
    threshold, integer shift of the monitored array.

    `emulate_gdt_byte_write`: the reference writes the shifted array with `to_raster(path, data)` whose band type
    defaults to `gdal.GDT_Byte` (core.py:781, image.py:388), so a uint16 monitored image comes back clamped to
    [0, 255] uint8 and KLT then runs on that.  Off by default (native dtype kept); switch it on to
    reproduce the reference's files bit for bit.

    Returns:
        (shifted monitored array, x_offset, y_offset) or None when no axis exceeds the threshold.
        The caller adds the offsets back to dx / dy (core.py:248-249) and skips ZNCC (core.py:876).
    """
    offsets = LargeOffsetMatcher(reference_image, monitored_image, ctx=ctx).match()
    logger.info("Large offset found: %s", offsets)
    if abs(offsets[1]) < bias_correction_min_threshold:
        offsets[1] = 0
    if abs(offsets[0]) < bias_correction_min_threshold:
        offsets[0] = 0
    if offsets[0] == 0 and offsets[1] == 0:
        return None
    shifted = ops.shift_image(np.asarray(monitored_image.array), x_off=offsets[1], y_off=offsets[0], ctx=ctx)
    if emulate_gdt_byte_write and shifted.dtype != np.uint8:
        shifted = np.clip(np.nan_to_num(shifted), 0, 255).astype(np.uint8)
    return shifted, float(offsets[1]), float(offsets[0])
