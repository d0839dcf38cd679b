"""KLT matcher -- drop-in for `karios.matcher.klt` with the numeric work on MI355X.

Same public surface as the reference module (`karios/matcher/klt.py`):
`KLT(conf, gen_laplacian=False, out_dir=None).match(mon, ref, mask)`, the properties
`auto_selected_ksize` / `auto_selected_polarity`, and the module function
`klt_tracker(ref_data, image_data, mask, conf, p0=None)`.  Where the reference calls
cv2 / numpy on whole tiles, this module calls `karios_amd.ops` (HIP kernels through the
C ABI).  Tile ordering, None conventions, DataFrame columns/dtypes and the float32
forward-backward arithmetic follow the reference line by line (cited inline).
"""
from __future__ import annotations

import itertools
import logging
import os
from collections import Counter
from collections.abc import Iterator

import numpy as np
from pandas import DataFrame

from .. import ops

logger = logging.getLogger(__name__)

LAPLACIAN_AUTO_CANDIDATES = [3, 5, 7, 9, 11]  # reference klt.py:39


def _to_uint8(arr: np.ndarray) -> np.ndarray:
    """Normalize an array to uint8, no-op if already uint8 (reference klt.py:42-49)."""
    if arr.dtype == np.uint8:
        return arr
    return ops.to_uint8(arr)


def _filter_outliers(x0, y0, x1, y1, score):
    """Iterative 3-sigma / 20 px clip (reference klt.py:52-71, `__filter_outliers`)."""
    dx = x1 - x0
    dy = y1 - y0
    while True:
        ind = (
            (np.abs(dx - dx.mean()) < 3 * dx.std())
            & (np.abs(dy - dy.mean()) < 3 * dy.std())
            & (np.abs(dx - dx.mean()) < 20)
            & (np.abs(dy - dy.mean()) < 20)
        )
        if int(np.count_nonzero(ind)) == len(dx):
            break
        dx, dy = dx[ind], dy[ind]
        x0, x1, y0, y1, score = x0[ind], x1[ind], y0[ind], y1[ind], score[ind]
    return x0, y0, x1, y1, score


def _frame_from_tracks(p0, p1, p0r, conf) -> tuple[DataFrame, int]:
    """Forward-backward test, score and DataFrame assembly (reference klt.py:142-170).
    LK status is deliberately ignored, as in the reference (klt.py:142-144)."""
    d = abs(p0 - p0r).reshape(-1, 2).max(-1)
    back_threshold = 0.1
    st = d < back_threshold
    ninit = len(p0)
    p0, p1, d = p0[st], p1[st], d[st]
    score = 1 - d / back_threshold
    x0 = p0[:, 0, 0].reshape(len(p0))
    y0 = p0[:, 0, 1].reshape(len(p0))
    x1 = p1[:, 0, 0].reshape(len(p1))
    y1 = p1[:, 0, 1].reshape(len(p1))
    if conf.outliers_filtering:
        logger.info("Filter outliers")
        x0, y0, x1, y1, score = _filter_outliers(x0, y0, x1, y1, score)
    frame = DataFrame.from_dict({"x0": x0, "y0": y0, "dx": x1 - x0, "dy": y1 - y0, "score": score})
    return frame, ninit


def _sorted_tile_frame(p0, p1, p0r, conf, x_off=0, y_off=0) -> tuple[DataFrame, int]:
    """`_frame_from_tracks` + tile offsets + `sort_values(by=["x0", "y0"])` (reference klt.py:341-348) built in
    one go: same rows, same values, same (permuted) index labels as the reference's in-place sort, without
    the pandas sort machinery (key pairs are unique, so the order is fully determined)."""
    frame, ninit = _frame_from_tracks(p0, p1, p0r, conf)
    x0 = frame["x0"].to_numpy() + x_off
    y0 = frame["y0"].to_numpy() + y_off
    order = np.lexsort((y0, x0))
    cols = {"x0": x0[order], "y0": y0[order]}
    for c in ("dx", "dy", "score"):
        cols[c] = frame[c].to_numpy()[order]
    return DataFrame(cols, index=order, copy=False), ninit


def klt_tracker(ref_data, image_data, mask, conf, p0=None, ctx=None) -> tuple[DataFrame, int] | None:
    """Run KLT (reference klt.py:83-172): Shi-Tomasi corners on `ref_data` (unless `p0` is
    given), pyramidal LK ref->image and image->ref, forward-backward filtering.

    Returns:
        (DataFrame[x0, y0, dx, dy, score] float32, Ninit) or None when no feature is extracted.
    """
    logger.info("Start tracking")
    tracks = ops.klt_track(ref_data, image_data, mask, conf, p0=p0, ctx=ctx)
    if tracks is None:
        logger.info("No features extracted")
        return None
    result = _frame_from_tracks(*tracks, conf)
    logger.info("Tracking finished")
    return result


class KLT:
    # pylint: disable=too-few-public-methods
    """Class to execute KLT matcher (reference klt.py:175-545)."""

    def __init__(self, conf, gen_laplacian: bool = False, out_dir: str | None = None, ctx=None):
        self._conf = conf
        self._gen_laplacian = gen_laplacian
        self._out_dir = out_dir
        self._ctx = ctx
        self._auto_selected_ksizes: list[tuple[int, int]] = []
        self._selected_polarities: list[str] = []

    # ------------------------------------------------------------------ tiling
    def tile_boxes(self, x_size: int, y_size: int) -> list[tuple[int, int, int, int]]:
        """(x_off, y_off, x_size, y_size) of every tile in the reference's order: x outer
        (skipping x_off < xStart), y inner, edge tiles clipped (klt.py:220-249)."""
        boxes = []
        ts = self._conf.tile_size
        for x_off in range(0, x_size, ts):
            if x_off < self._conf.xStart:
                continue
            for y_off in range(0, y_size, ts):
                bx = ts if x_off + ts < x_size else x_size - x_off
                by = ts if y_off + ts < y_size else y_size - y_off
                boxes.append((x_off, y_off, bx, by))
        return boxes

    def match(self, mon_img, ref_img, mask) -> Iterator[DataFrame]:
        """Run KLT on the image to monitor against a reference image (klt.py:198-234).

        Yields one DataFrame per tile that produced points."""
        logger.info("KLT...")
        logger.info("%s %s", mon_img.x_size, mon_img.y_size)
        self._log_polarity_setting()
        for x_off, y_off, _, _ in self.tile_boxes(mon_img.x_size, mon_img.y_size):
            points = self._match_tile(x_off, y_off, mon_img, ref_img, mask)
            if points is None:
                continue
            yield points
        self._log_polarity_summary()

    def _match_tile(self, x_off, y_off, mon_img, ref_img, mask) -> DataFrame | None:
        logger.info("Tile: %s %s (%s %s)", x_off, y_off, mon_img.x_size, mon_img.y_size)
        ts = self._conf.tile_size
        x_size = ts if x_off + ts < mon_img.x_size else mon_img.x_size - x_off
        y_size = ts if y_off + ts < mon_img.y_size else mon_img.y_size - y_off

        ref_box = ref_img.read(1, x_off, y_off, x_size, y_size)
        img_box = mon_img.read(1, x_off, y_off, x_size, y_size)
        mask_box = mask.read(1, x_off, y_off, x_size, y_size) if mask else None
        nodata = (getattr(mon_img, "no_data_value", None), getattr(ref_img, "no_data_value", None))

        polarity_mode = self._conf.laplacian_invert_polarity
        ksize = self._conf.laplacian_kernel_size
        fused = polarity_mode != "auto" and ksize != "auto" and not self._gen_laplacian
        if fused:
            # whole tile in one device pipeline (stretch, Laplacians, mask, GFTT, LK)
            mon_k, ref_k = self._resolve_ksize(ksize)
            status, tracks = ops.klt_tile(ref_box, img_box, self._conf, mask_box=mask_box, nodata_ref=nodata[1],
                                          nodata_mon=nodata[0], mon_ksize=mon_k, ref_ksize=ref_k,
                                          invert_mon=bool(polarity_mode), ctx=self._ctx)
            if status == "no_valid_pixels":
                logger.info("-- No valid pixels, skipping this tile")
                return None
            results = None if tracks is None else _frame_from_tracks(*tracks, self._conf)
            dump = None
        else:
            if mask_box is None:
                mask_box, valid_pixels = ops.auto_mask(img_box, ref_box, nodata[0], nodata[1], ctx=self._ctx)
            else:
                valid_pixels = int(np.count_nonzero(np.asarray(mask_box) > 0))
            if valid_pixels == 0:
                logger.info("-- No valid pixels, skipping this tile")
                return None
            logger.info("Nb valid pixels: %s/%s", valid_pixels, x_size * y_size)
            if polarity_mode == "auto":
                normal_res, normal_dump = self._laplacian_track_once(img_box, ref_box, mask_box, invert_mon=False)
                inverted_res, inverted_dump = self._laplacian_track_once(img_box, ref_box, mask_box, invert_mon=True)
                results, dump = self._select_best_polarity(normal_res, normal_dump, inverted_res, inverted_dump)
            else:
                results, dump = self._laplacian_track_once(img_box, ref_box, mask_box, invert_mon=bool(polarity_mode))

        if dump is not None:
            img_lap, ref_lap, mon_ksize, ref_ksize, invert_mon = dump
            if self._conf.laplacian_kernel_size == "auto":
                self._auto_selected_ksizes.append((mon_ksize, ref_ksize))
            if self._gen_laplacian:
                suffix = "_inv" if invert_mon else ""
                self._write_laplacian(f"mon_laplacian{suffix}_k{mon_ksize}_{x_off}_{y_off}_{x_size}_{y_size}", img_lap)
                self._write_laplacian(f"ref_laplacian_k{ref_ksize}_{x_off}_{y_off}_{x_size}_{y_size}", ref_lap)

        if not results:
            logger.warning("No result for tile %s %s (%s %s)", x_off, y_off, mon_img.x_size, mon_img.y_size)
            return None

        points, initial_nb_points = results
        points["x0"] = points["x0"] + x_off
        points["y0"] = points["y0"] + y_off
        logger.info("NbPoints(init/final): %s / %s", initial_nb_points, len(points.dx))
        logger.info("DX/DY(KLT) MEAN: %s / %s", points.dx.mean(), points.dy.mean())
        logger.info("DX/DY(KLT) STD: %s / %s", points.dx.std(), points.dy.std())
        points.sort_values(by=["x0", "y0"], inplace=True)
        return points

    def _write_laplacian(self, stem: str, lap: np.ndarray) -> None:
        """Debug dump of a Laplacian (reference uses skimage.io.imsave, klt.py:307-322)."""
        path = os.path.join(self._out_dir or ".", stem)
        try:
            from skimage import io  # type: ignore
            io.imsave(path + ".tif", lap)
        except ImportError:
            np.save(path + ".npy", lap)

    # ------------------------------------------------------------------ properties
    @property
    def auto_selected_ksize(self) -> tuple[int, int] | None:
        """Most common (mon_ksize, ref_ksize) pair across auto-mode tiles (klt.py:351-356)."""
        if not self._auto_selected_ksizes:
            return None
        return Counter(self._auto_selected_ksizes).most_common(1)[0][0]

    @property
    def auto_selected_polarity(self) -> str | None:
        """Most common polarity ('normal' / 'inverted') across tiles in 'auto' mode (klt.py:399-405)."""
        if not self._selected_polarities:
            return None
        return Counter(self._selected_polarities).most_common(1)[0][0]

    # ------------------------------------------------------------------ unfused path
    @staticmethod
    def _resolve_ksize(ksize):
        """int | {"mon","ref"} -> (mon_ksize, ref_ksize) (klt.py:431-432)."""
        if isinstance(ksize, dict):
            return ksize.get("mon", ksize.get("ref", 1)), ksize.get("ref", ksize.get("mon", 1))
        return ksize, ksize

    def _apply_laplacian_and_track(self, img_box, ref_box, mask_box, mon_ksize, ref_ksize):
        lap_img = ops.laplacian_u8(_to_uint8(img_box), mon_ksize, ctx=self._ctx)
        lap_ref = ops.laplacian_u8(_to_uint8(ref_box), ref_ksize, ctx=self._ctx)
        return klt_tracker(lap_ref, lap_img, mask_box, self._conf, ctx=self._ctx)

    def _laplacian_track_once(self, img_box, ref_box, mask_box, invert_mon: bool):
        """Laplacian + KLT once (klt.py:407-436) -> (result, (img_lap, ref_lap, mon_k, ref_k, invert) | None)."""
        img_for_lap = ops.to_uint8(img_box, invert=True, ctx=self._ctx) if invert_mon else img_box
        ksize = self._conf.laplacian_kernel_size
        if ksize == "auto":
            result, _, best_ksize = self._match_tile_auto_ksize(img_for_lap, ref_box, mask_box)
            if best_ksize is None:
                return result, None
            mon_ksize, ref_ksize = best_ksize
            img_lap = ops.laplacian_u8(_to_uint8(img_for_lap), mon_ksize, ctx=self._ctx)
            ref_lap = ops.laplacian_u8(_to_uint8(ref_box), ref_ksize, ctx=self._ctx)
            return result, (img_lap, ref_lap, mon_ksize, ref_ksize, invert_mon)
        mon_ksize, ref_ksize = self._resolve_ksize(ksize)
        img_lap = ops.laplacian_u8(_to_uint8(img_for_lap), mon_ksize, ctx=self._ctx)
        ref_lap = ops.laplacian_u8(_to_uint8(ref_box), ref_ksize, ctx=self._ctx)
        result = klt_tracker(ref_lap, img_lap, mask_box, self._conf, ctx=self._ctx)
        return result, (img_lap, ref_lap, mon_ksize, ref_ksize, invert_mon)

    def _select_best_polarity(self, normal_res, normal_dump, inverted_res, inverted_dump):
        """Keep the polarity with the higher inlier ratio, 'normal' first on ties (klt.py:438-463)."""
        candidates = []
        for label, res, dump in (("normal", normal_res, normal_dump), ("inverted", inverted_res, inverted_dump)):
            if res is None:
                continue
            points, ninit = res
            ratio = len(points) / ninit if ninit > 0 else 0.0
            candidates.append((label, ratio, res, dump))
        if not candidates:
            logger.info("Auto polarity: no candidate produced a result")
            return None, None
        candidates.sort(key=lambda c: c[1], reverse=True)
        label, ratio, result, dump = candidates[0]
        self._selected_polarities.append(label)
        logger.info("Auto polarity selected: %s (inlier ratio=%.3f)", label, ratio)
        return result, dump

    def _match_tile_auto_ksize(self, img_box, ref_box, mask_box):
        """Try the 25 (mon_ksize, ref_ksize) pairs, keep the highest inlier ratio, first wins
        ties (klt.py:465-545).  Returns (best result | None, {pair: ratio}, best pair | None)."""
        combinations = list(itertools.product(LAPLACIAN_AUTO_CANDIDATES, repeat=2))
        if not getattr(self._conf, "outliers_filtering", False):
            # batched search on the device (km_klt_auto_ksize_frame_dev): the tile is uploaded once, the 10 Laplacians,
            # their pyramids and the 5 corner lists are shared by the 25 tracker runs
            from ..resident import ResidentPair
            pair = ResidentPair.upload(np.ascontiguousarray(img_box), np.ascontiguousarray(ref_box), mask_box, ctx=self._ctx)
            frame, scores, best_ksize, ninit = pair.match_tile_auto_ksize(self._conf, candidates=LAPLACIAN_AUTO_CANDIDATES)
            for (mk, rk) in combinations:
                logger.info("Auto laplacian: mon_ksize=%s ref_ksize=%s inlier ratio=%.3f", mk, rk, scores[(mk, rk)])
            logger.info("Auto laplacian selected: mon_ksize=%s ref_ksize=%s", *(best_ksize if best_ksize else (None, None)))
            return (None if frame is None else (frame, ninit)), scores, best_ksize
        img_uint8 = _to_uint8(img_box)
        ref_uint8 = _to_uint8(ref_box)
        mon_laplacians = {k: ops.laplacian_u8(img_uint8, k, ctx=self._ctx) for k in LAPLACIAN_AUTO_CANDIDATES}
        ref_laplacians = {k: ops.laplacian_u8(ref_uint8, k, ctx=self._ctx) for k in LAPLACIAN_AUTO_CANDIDATES}
        ref_p0s = {
            k: ops.good_features_to_track(lap, self._conf.maxCorners, self._conf.qualityLevel, self._conf.minDistance,
                                          mask=mask_box, blockSize=self._conf.blocksize, ctx=self._ctx)
            for k, lap in ref_laplacians.items()
        }
        scores: dict[tuple[int, int], float] = {}
        best_result, best_ratio, best_ksize = None, -1.0, None
        # The reference maps these runs over a ThreadPoolExecutor and consumes them in submission
        # order; one GPU stream runs them back to back in the same order, same tie rule.
        for mon_ksize, ref_ksize in combinations:
            logger.info("Auto laplacian: trying mon_ksize=%s ref_ksize=%s", mon_ksize, ref_ksize)
            p0 = ref_p0s[ref_ksize]
            result = None
            if p0 is not None:
                result = klt_tracker(ref_laplacians[ref_ksize], mon_laplacians[mon_ksize], mask_box, self._conf, p0=p0,
                                     ctx=self._ctx)
            if result is None:
                scores[(mon_ksize, ref_ksize)] = 0.0
                continue
            points, ninit = result
            ratio = len(points) / ninit if ninit > 0 else 0.0
            scores[(mon_ksize, ref_ksize)] = ratio
            if ratio > best_ratio:
                best_ratio, best_result, best_ksize = ratio, result, (mon_ksize, ref_ksize)
        logger.info("Auto laplacian selected: mon_ksize=%s ref_ksize=%s (inlier ratio=%.3f)",
                    best_ksize[0] if best_ksize else None, best_ksize[1] if best_ksize else None, best_ratio)
        return best_result, scores, best_ksize

    # ------------------------------------------------------------------ logging
    def _log_polarity_setting(self) -> None:
        mode = self._conf.laplacian_invert_polarity
        if mode == "auto":
            logger.info("Laplacian polarity: 'auto' - each tile runs twice, the higher inlier ratio is kept")
        elif mode:
            logger.info("Laplacian polarity: 'inverted' - monitored pixels are inverted (255 - pixel) before Laplacian")
        else:
            logger.info("Laplacian polarity: 'normal' - no inversion before Laplacian")

    def _log_polarity_summary(self) -> None:
        if self._conf.laplacian_invert_polarity != "auto":
            return
        if not self._selected_polarities:
            logger.info("Auto polarity: no tile produced a result")
            return
        counts = Counter(self._selected_polarities)
        total = sum(counts.values())
        dominant, dominant_count = counts.most_common(1)[0]
        details = ", ".join(f"{name}={n}/{total}" for name, n in counts.most_common())
        logger.info("Auto polarity dominant choice: '%s' (%d/%d tiles, %s)", dominant, dominant_count, total, details)
