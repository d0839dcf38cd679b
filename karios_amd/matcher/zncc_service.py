"""ZNCC / NMI confidence scores of the matched key points on MI355X, behind `karios.matcher.zncc_service`.

Contract of the reference (`karios/matcher/zncc_service.py:45-126, 154-297`): for every row of a key-point frame the 43x43
windows around `(int(x0), int(y0))` in the reference image and `(round(x0 + dx), round(y0 + dy))` in the monitored image are
correlated (population statistics, float64); rows whose 57x57 chip would leave either image score NaN, so do windows
without variance; the result is a Series on the frame's index.  The reference walks the frame with `DataFrame.apply`
(~160 us per key point); here the whole frame is one kernel launch, one wavefront per key point.
"""
from __future__ import annotations

import logging

import numpy as np
from pandas import DataFrame, Series

from .. import ops
from .._lib import _DTYPES
from ..resident import keep_shared_across, shared_pair

logger = logging.getLogger(__name__)

WINDOW_HALF = 21      # correlated window: 43 x 43
CHIP_SIZE = 57        # the bounds rule is stated on the 57 x 57 chip the reference cuts first


def _zncc2(img1, img2, u1: int, v1: int, u2: int, v2: int, n: int) -> float:
    """ZNCC of the (2n+1)^2 windows centred at (row u1, column v1) of `img1` and (u2, v2) of `img2`: ValueError for a negative
    half-size, IndexError when a window leaves its image, NaN when a window has no variance.  Any n, any numeric dtypes."""
    if n < 0:
        raise ValueError("Window half-size n must be non-negative")
    for img, (row, col) in ((img1, (u1, v1)), (img2, (u2, v2))):
        rows, cols = np.shape(img)
        if not (n <= row < rows - n and n <= col < cols - n):
            raise IndexError("Patch window extends beyond image boundaries")
    values, _ = ops.zncc_windows(img1, img2, [u1], [v1], [u2], [v2], n)
    return float(values[0])


def _frame_arithmetic_dtype(df: DataFrame):
    """`DataFrame.apply(axis=1)` hands the reference's row function a Series of the frame's COMMON dtype, so `x0 + dx` is a
    float32 sum for the all-float32 frames of `KLT.match` and a float64 sum as soon as the frame carries a float64 column."""
    kinds = [np.dtype(t) for t in df.dtypes if np.issubdtype(t, np.number)]
    return np.result_type(*kinds) if kinds else np.dtype(np.float64)


def _chip_centres(df: DataFrame, margin: int, ref_shape, mon_shape):
    """Window centres of every row and the rows the reference scores at all.
    -> (rows u_ref, columns v_ref, u_mon, v_mon as int64 arrays, boolean `inside`)."""
    dt = _frame_arithmetic_dtype(df)
    x0, y0, dx, dy = (df[c].to_numpy().astype(dt, copy=False) for c in ("x0", "y0", "dx", "dy"))
    with np.errstate(invalid="ignore"):
        finite = np.isfinite(x0) & np.isfinite(y0) & np.isfinite(x0 + dx) & np.isfinite(y0 + dy)
        cx0 = np.where(finite, x0, 0).astype(np.int64)                    # int(): truncation
        cy0 = np.where(finite, y0, 0).astype(np.int64)
        cx1 = np.rint(np.where(finite, x0 + dx, 0)).astype(np.int64)      # round(): half to even, on the sum in the frame's dtype
        cy1 = np.rint(np.where(finite, y0 + dy, 0)).astype(np.int64)
    inside = finite & (np.minimum(np.minimum(cx0, cy0), np.minimum(cx1, cy1)) >= margin)
    inside &= (cx0 < ref_shape[1] - margin) & (cy0 < ref_shape[0] - margin) & (cx1 < mon_shape[1] - margin) & (cy1 < mon_shape[0] - margin)
    return cy0, cx0, cy1, cx1, inside


def _kernel_ready(df: DataFrame, ref: np.ndarray, mon: np.ndarray) -> bool:
    """The fused 43x43 kernel applies the float32 rounding rule itself and reads one pixel type for both images."""
    return ref.dtype == mon.dtype and ref.dtype in _DTYPES and _frame_arithmetic_dtype(df) == np.float32


class ZNCCService:
    """Scores the key points of a frame against the raw (full-resolution) images."""

    def __init__(self, ctx=None):
        self._chip_size = CHIP_SIZE
        self._chip_margin = (CHIP_SIZE - 1) // 2
        self._ctx = ctx

    def compute_zncc(self, df: DataFrame, monitored, reference) -> Series:
        """ZNCC per row of `df` (columns x0, y0, dx, dy); NaN where the reference skips the row.  Index = `df.index`."""
        values = self._score(df, monitored, reference, self._zncc_values)
        logger.info("ZNCC: %d key points scored, %d without a value", len(df), int(np.isnan(values).sum()) if len(df) else 0)
        return Series(values, index=df.index, dtype=np.float64)

    def compute_mi(self, df: DataFrame, monitored, reference) -> Series:
        """Normalised mutual information 2*MI/(H(X)+H(Y)) of the 57x57 chips (32 bins) per row (zncc_service.py:240-287)."""
        values = self._score(df, monitored, reference, self._nmi_values)
        logger.info("NMI: %d key points scored", len(df))
        return Series(values, index=df.index, dtype=np.float64)

    # ------------------------------------------------------------------ internals
    def _score(self, df, monitored, reference, scorer) -> np.ndarray:
        try:
            if len(df) == 0:
                return np.empty(0, np.float64)
            self._rasters = (monitored, reference)
            return scorer(df, np.asarray(reference.array), np.asarray(monitored.array))
        finally:
            self._rasters = ()
            with keep_shared_across(monitored, reference):      # (this call is the service's own: nothing was edited since the upload)
                monitored.clear_cache()      # the reference drops both rasters from its cache after scoring (:179-180)
                reference.clear_cache()

    def _zncc_values(self, df, ref, mon) -> np.ndarray:
        if _kernel_ready(df, ref, mon):
            cols = [df[c].to_numpy(dtype=np.float32, copy=False) for c in ("x0", "y0", "dx", "dy")]
            if ref.shape == mon.shape:
                return shared_pair(mon, ref, self._ctx, rasters=getattr(self, "_rasters", ())).zncc(*cols)      # resident already when KLT.match just ran on them
            return ops.zncc_batch(ref, mon, *cols, ctx=self._ctx)
        # cross-sensor pairs / other pixel types / float64 frames: centres on the host, windows by the generic kernel
        u0, v0, u1, v1, inside = _chip_centres(df, self._chip_margin, ref.shape, mon.shape)
        out = np.full(len(df), np.nan)
        if inside.any():
            out[inside], _ = ops.zncc_windows(ref, mon, u0[inside], v0[inside], u1[inside], v1[inside], WINDOW_HALF, ctx=self._ctx)
        return out

    def _nmi_values(self, df, ref, mon) -> np.ndarray:
        if not _kernel_ready(df, ref, mon):
            ref, mon = _common_pixel_type(ref, mon)
        return _mi_scores(df, ref, mon, self._ctx, rasters=getattr(self, "_rasters", ()))[1]

    def _extract_chip(self, x: int, y: int, image):
        m = self._chip_margin
        return image.array[y - m:y + m + 1, x - m:x + m + 1]


def _mi_scores(df: DataFrame, ref: np.ndarray, mon: np.ndarray, ctx, rasters=()):
    """(Studholme, NMI) of every row; images of equal shape go through the shared resident pair (one upload for all services,
    one kernel run for both scores)."""
    cols = [df[c].to_numpy(dtype=np.float32, copy=False) for c in ("x0", "y0", "dx", "dy")]
    if ref.shape == mon.shape and ref.dtype == mon.dtype:
        return shared_pair(mon, ref, ctx, rasters=rasters).mutual_info(*cols)
    return ops.mi_batch(ref, mon, *cols, ctx=ctx)


def _common_pixel_type(ref: np.ndarray, mon: np.ndarray):
    """Both images in ONE pixel type the histogram kernel reads, without changing a value (the scores only see values)."""
    for target in (np.uint8, np.uint16, np.int16, np.float32):
        if all(a.dtype == target or np.can_cast(a.dtype, target, "safe") for a in (ref, mon)):
            return ref.astype(target, copy=False), mon.astype(target, copy=False)
    if all(np.array_equal(a, a.astype(np.float32)) for a in (ref, mon)):     # e.g. uint16 next to int16, small int32 counts
        return ref.astype(np.float32), mon.astype(np.float32)
    raise ops.KariosHipError(f"mutual information: pixel types {ref.dtype} / {mon.dtype} have no common type the device reads exactly")
