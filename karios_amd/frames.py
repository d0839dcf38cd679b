"""Key-point frames: from the tracker's three point lists to the DataFrame `KLT.match` yields.

What the reference computes between `calcOpticalFlowPyrLK` and its return (`karios/matcher/klt.py:142-170, 341-348`):
forward-backward distance `d = max(|p0 - p0r|)` per point in float32, keep `d < float32(0.1)` (the LK status is ignored),
`score = 1 - d / 0.1`, `dx, dy = p1 - p0`, optionally an iterative sigma clip, the tile offset added to `x0, y0`, rows
ordered by `(x0, y0)` with the labels pandas' in-place sort leaves behind (the row's position before the sort).

On the fixed-parameter path all of this happens on the device (k_frame.hip) and only `block_to_frame` runs here; the
functions below serve the host-buffer entry points (`klt_tracker`) and the sigma-clip variant, whose float32 means and
standard deviations must be numpy's.
"""
from __future__ import annotations

import numpy as np
from pandas import DataFrame

FB_LIMIT = np.float32(0.1)     # forward-backward tolerance in pixels (klt.py:143)
COLUMNS = ("x0", "y0", "dx", "dy", "score")


def track_columns(p0, p1, p0r) -> tuple[dict[str, np.ndarray], int]:
    """(N,1,2) float32 point lists -> float32 columns of the tracks that pass the forward-backward test, in corner order,
    and N (the reference's `Ninit`)."""
    a = np.asarray(p0, np.float32).reshape(-1, 2)
    b = np.asarray(p1, np.float32).reshape(-1, 2)
    back = np.asarray(p0r, np.float32).reshape(-1, 2)
    gap = np.abs(a - back).max(axis=1)
    keep = np.flatnonzero(gap < FB_LIMIT)          # NaN distances compare false: dropped, like numpy in the reference
    a, b, gap = a[keep], b[keep], gap[keep]
    cols = {"x0": a[:, 0].copy(), "y0": a[:, 1].copy(), "dx": b[:, 0] - a[:, 0], "dy": b[:, 1] - a[:, 1],
            "score": 1 - gap / FB_LIMIT}
    return cols, len(np.asarray(p0).reshape(-1, 2))


def sigma_clip(dx: np.ndarray, dy: np.ndarray, n_sigma: float = 3.0, limit: float = 20.0) -> np.ndarray:
    """Indices of the displacements that survive the reference's outlier loop (klt.py:52-71): drop every point further than
    `n_sigma` population standard deviations or `limit` pixels from the mean displacement (either axis), recompute mean and
    deviation on the survivors, repeat until a pass drops nothing.  The statistics are taken on the COMPACTED survivor
    arrays each round - float32 pairwise sums depend on the array they run over."""
    alive = np.arange(len(dx))
    while len(alive):
        u, v = dx[alive], dy[alive]
        off_u, off_v = np.abs(u - u.mean()), np.abs(v - v.mean())
        ok = (off_u < n_sigma * u.std()) & (off_v < n_sigma * v.std()) & (off_u < limit) & (off_v < limit)
        if ok.all():
            break
        alive = alive[ok]
    return alive


def assemble(cols: dict[str, np.ndarray], x_off=0, y_off=0, clip_outliers: bool = False, ordered: bool = True) -> DataFrame:
    """Columns of `track_columns` -> frame.  `ordered`: rows by (x0, y0), index = position before the ordering (what
    `sort_values(inplace=True)` leaves); otherwise corner order with a fresh RangeIndex (what `klt_tracker` returns)."""
    if clip_outliers:
        keep = sigma_clip(cols["dx"], cols["dy"])
        cols = {k: v[keep] for k, v in cols.items()}
    x0, y0 = cols["x0"] + x_off, cols["y0"] + y_off
    if x0.dtype != np.float32:                     # python / numpy integer offsets keep float32; anything else is cast back
        x0, y0 = x0.astype(np.float32), y0.astype(np.float32)
    if not ordered:
        return DataFrame({"x0": x0, "y0": y0, "dx": cols["dx"], "dy": cols["dy"], "score": cols["score"]})
    order = np.lexsort((y0, x0))                   # key pairs are unique (integer corners): the order is total
    data = {"x0": x0[order], "y0": y0[order]}
    data.update({k: cols[k][order] for k in ("dx", "dy", "score")})
    return DataFrame(data, index=order, copy=False)


SCORE_COLUMNS = ("zncc_score", "mutual_info_score", "mi_score")     # float64 columns of a scored frame, in block order (core.py:894-907)


def block_words(cap: int, with_zncc=False) -> int:
    """float32 words of one frame block.  `with_zncc` counts the float64 score columns behind the six float32 ones: False / 0 none,
    True / 1 `zncc_score`, 3 `zncc_score | mutual_info_score | mi_score` (the whole of `_handle_klt_results`' scoring)."""
    return 4 + (6 + 2 * int(with_zncc)) * cap


def block_to_frame(block: np.ndarray, cap: int, with_zncc=False, radial: bool = False, own: bool = False) -> DataFrame | None:
    """Frame block of the device pipeline (km_klt_tile_frame[_zncc]_dev: 4 int32 {rows, Ninit, flags, candidates}, then `cap` float32
    per column x0 | y0 | dx | dy | score | index bits, then `cap` float64 per score column - `with_zncc` = their number, see
    `block_words`) -> DataFrame; None when no corner was found.
    `radial`: with the `radial error` / `angle` columns of `radial_angle_columns` already in place (one DataFrame construction
    instead of two column insertions, which cost more than every array operation of the host stage together).  `own`: the block
    belongs to the frame from now on (the columns become views of it instead of copies)."""
    rows, n_init = (int(v) for v in block[:2].view(np.int32))
    if n_init == 0:
        return None
    body = block[4:]
    take = (lambda a: a) if own else (lambda a: a.copy())
    data = {name: take(body[i * cap:i * cap + rows]) for i, name in enumerate(COLUMNS)}
    for k in range(int(with_zncc)):
        data[SCORE_COLUMNS[k]] = take(body[(6 + 2 * k) * cap:(8 + 2 * k) * cap].view(np.float64)[:rows])
    if radial:
        data["radial error"], data["angle"] = _radial_angle(data["dx"], data["dy"])
    labels = body[5 * cap:5 * cap + rows].view(np.int32).astype(np.int64)
    return DataFrame(data, index=labels, copy=False)


def _radial_angle(dx: np.ndarray, dy: np.ndarray):
    """|(dx, dy)| and atan2(dy, dx) in degrees, float32 like the displacement columns they are computed from (core.py:872-873).
    numpy on the host: a device atan2 would not reproduce numpy's last bit."""
    return np.sqrt(dx ** 2 + dy ** 2), np.degrees(np.arctan2(dy, dx))


def radial_angle_columns(frame: DataFrame) -> DataFrame:
    """Adds `radial error` = |(dx, dy)| and `angle` = atan2(dy, dx) in degrees, float32 like the displacement columns they are
    computed from (core.py:872-873).  numpy on the host: a device atan2 would not reproduce numpy's last bit."""
    if "radial error" in frame.columns and "angle" in frame.columns:      # built with block_to_frame(..., radial=True)
        return frame
    frame["radial error"], frame["angle"] = _radial_angle(frame["dx"].to_numpy(), frame["dy"].to_numpy())
    return frame
