"""Post-matching step of the reference's API layer on resident data:

  * `handle_klt_results`   - `KariosAPI._handle_klt_results` (karios/api/core.py:848-921): radial error / angle columns,
                              ZNCC + both mutual-information scores of the rows with score >= confidence threshold,
                              CSV written tile by tile (`sep=";"`, header once, no index), frames concatenated;
  * `filter_by_dn_values`  - `KariosAPI._filter_by_dn_values` (core.py:650-737): drop the key points under which the
                              reference or monitored image holds one of the excluded DN values / its no-data value.

The pixel work (ZNCC, MI / NMI, DN gather) runs on the device through `ResidentPair`; the column arithmetic and the
CSV formatting are the reference's own numpy / pandas expressions.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path
from typing import Iterable

import numpy as np
import pandas as pd

from ._lib import KariosHipError
from .frames import radial_angle_columns
from .resident import ResidentPair

CSV_COLUMNS = ["x0", "y0", "dx", "dy", "score", "radial error", "angle", "zncc_score", "mutual_info_score", "mi_score"]


def handle_klt_results(results: Iterable[pd.DataFrame], csv_file, pair: ResidentPair, confidence_threshold: float = 0.4,
                       large_shift_applied: bool = False) -> pd.DataFrame:
    """`KariosAPI._handle_klt_results` (core.py:848-921) for the frames of `KLT.match` / `ResidentPair.match*`.
    With `large_shift_applied` the scores are skipped like in the reference (core.py:909-910) and the CSV has 7 columns."""
    csv_file = Path(csv_file)
    all_frame = pd.DataFrame()
    for frame in results:
        if large_shift_applied:
            frame = radial_angle_columns(frame)
            if "zncc_score" in frame.columns:
                frame = frame.drop(columns=["zncc_score"])
        else:
            frame = pair.score_frame(frame, confidence_threshold, mutual_info=True)
            frame = frame[CSV_COLUMNS]           # the reference's column order, whatever produced zncc_score first
        if not csv_file.exists():
            frame.to_csv(csv_file, sep=";", index=False)
        else:
            frame.to_csv(csv_file, mode="a", sep=";", index=False, header=False)
        all_frame = pd.concat([all_frame, frame])
    return all_frame


def filter_by_dn_values(points: pd.DataFrame, pair: ResidentPair, no_values=None) -> pd.DataFrame:
    """`KariosAPI._filter_by_dn_values` (core.py:650-737) with the DN gather on the device.  `no_values` apply to both
    images; each image's own no-data value (pair.no_data_ref / no_data_mon) applies to that image only."""
    ref_nd, mon_nd = pair.no_data_ref, pair.no_data_mon
    if not no_values and ref_nd is None and mon_nd is None:
        return points
    n = len(points)
    if n == 0:
        return points[np.ones(0, bool)].copy()
    c = pair.ctx
    pair._ready()
    x0 = np.ascontiguousarray(points["x0"].to_numpy(), np.float32)
    y0 = np.ascontiguousarray(points["y0"].to_numpy(), np.float32)
    nv = np.ascontiguousarray([float(v) for v in (no_values or [])], np.float64)
    keep = np.empty(n, np.uint8)
    nr = C.byref(C.c_double(float(ref_nd))) if ref_nd is not None else None
    nm = C.byref(C.c_double(float(mon_nd))) if mon_nd is not None else None
    rc = c.lib.km_dn_keep_dev(c.handle, C.c_void_p(pair.ref_ptr), C.c_void_p(pair.mon_ptr), pair.code, pair.y_size, pair.x_size,
                              pair.x_size, pair.x_size, x0.ctypes.data_as(C.c_void_p), y0.ctypes.data_as(C.c_void_p), n,
                              nv.ctypes.data_as(C.c_void_p) if len(nv) else None, len(nv), nr, nm, keep.ctypes.data_as(C.c_void_p))
    if rc != 0:
        raw = c.lib.km_last_error(c.handle)
        msg = raw.decode() if raw else ""
        if "outside" in msg:
            raise IndexError(msg)                # numpy's fancy indexing raises for an out-of-bounds key point
        raise KariosHipError(f"km_dn_keep_dev: {msg}")
    return points[keep.astype(bool)].copy()
