"""KLT matcher on MI355X behind the `karios.matcher.klt` surface.

Public names and contracts are the reference's (`karios/matcher/klt.py:83-234, 351-405`): `KLT(conf, gen_laplacian, out_dir)`,
`KLT.match(mon_img, ref_img, mask)` yielding one float32 `x0, y0, dx, dy, score` frame per productive tile in x-outer / y-inner
order, the `auto_selected_ksize` / `auto_selected_polarity` properties and `klt_tracker(ref, img, mask, conf, p0)`.

Everything behind them is organised around the device instead of around cv2 calls: a tile's two boxes are uploaded ONCE
(`ResidentPair`), every trial the configuration asks for - fixed parameters, the 5x5 Laplacian kernel-size search, both
polarities - runs on that resident copy (stretch, Laplacians, mask, corners, LK forward / backward, forward-backward test and
the (x0, y0) ordering in libkarios_hip), and only finished frames and inlier counts come back; the host compares trials and
keeps tallies.
"""
from __future__ import annotations

import logging
import os
from collections import Counter
from collections.abc import Iterator
from dataclasses import dataclass

import numpy as np
from pandas import DataFrame

from .. import frames, ops, tiling
from ..resident import ResidentPair, shared_pair

logger = logging.getLogger(__name__)

LAPLACIAN_AUTO_CANDIDATES = list(tiling.AUTO_KSIZE_CANDIDATES)
_DEVICE_DTYPES = (np.uint8, np.uint16, np.int16, np.float32)


def klt_tracker(ref_data, image_data, mask, conf, p0=None, ctx=None) -> tuple[DataFrame, int] | None:
    """Corners of `ref_data` (or the given `p0`), tracked into `image_data` and back; the tracks whose return trip ends
    within 0.1 px of their start are scored (reference klt.py:83-172; the LK status flags are ignored there too).

    Returns (frame[x0, y0, dx, dy, score] float32 in corner order, number of corners), or None without corners."""
    tracks = ops.klt_track(ref_data, image_data, mask, conf, p0=p0, ctx=ctx)
    if tracks is None:
        logger.info("klt_tracker: no corner found")
        return None
    cols, n_init = frames.track_columns(*tracks)
    frame = frames.assemble(cols, clip_outliers=bool(getattr(conf, "outliers_filtering", False)), ordered=False)
    logger.info("klt_tracker: %d of %d corners survive the forward-backward test", len(frame), n_init)
    return frame, n_init


@dataclass
class _Trial:
    """Outcome of one (polarity, kernel sizes) attempt on a tile."""
    frame: DataFrame          # ordered tile frame, offsets applied
    n_init: int               # corners the tracker started from
    ksizes: tuple[int, int]   # (mon, ref) Laplacian kernel sizes used
    inverted: bool

    @property
    def inlier_ratio(self) -> float:
        return len(self.frame) / self.n_init if self.n_init > 0 else 0.0

    @property
    def polarity(self) -> str:
        return "inverted" if self.inverted else "normal"


class _TileSession:
    """A tile resident on the device for the duration of its trials."""

    def __init__(self, conf, tile: tiling.Tile, mon_box, ref_box, mask_box, nodata_mon, nodata_ref, ctx, rasters=()):
        self.conf, self.tile = conf, tile
        mon_box, ref_box = np.asarray(mon_box), np.asarray(ref_box)
        if mask_box is not None:
            mask_box = np.ascontiguousarray(mask_box, np.uint8)
        if mon_box.dtype != ref_box.dtype or mon_box.dtype.type not in _DEVICE_DTYPES:
            # Cross-sensor pairs (a uint8 chip against uint16 data) and pixel types the kernels do not read (int32, float64 ...):
            # the reference stretches each image on its own and derives the mask from the raw values, so do exactly that
            # up front and hand the device a uint8 pair with an explicit mask.
            if mask_box is None:
                mask_box = _validity_mask(mon_box, ref_box, nodata_mon, nodata_ref)
            mon_box, ref_box = _stretch_any(mon_box, ctx), _stretch_any(ref_box, ctx)
            nodata_mon = nodata_ref = None
        self.pair = ResidentPair.upload(mon_box, ref_box, mask_box, ctx=ctx, no_data_mon=nodata_mon, no_data_ref=nodata_ref)
        shared_pair(mon_box, ref_box, ctx, publish=self.pair, rasters=rasters)      # the scoring services find a whole-image tile resident
        self._host = (mon_box, ref_box)
        self.valid_pixels = -1                     # unknown until a fixed-parameter trial has run

    def fixed(self, ksizes: tuple[int, int], inverted: bool) -> _Trial | None:
        frame = self.pair.match_tile(self.conf, ksizes=ksizes, invert_mon=inverted, origin=self.tile[:2])
        self.valid_pixels = int(self.pair.ctx.stats().valid_pixels)
        if frame is None:
            return None
        return _Trial(frame, int(frame.attrs.get("Ninit", len(frame))), ksizes, inverted)

    def kernel_search(self, inverted: bool) -> _Trial | None:
        """Best of the 25 (mon, ref) kernel-size pairs by inlier ratio; the first pair in (mon outer, ref inner) order keeps a
        tie (klt.py:534-539)."""
        if not getattr(self.conf, "outliers_filtering", False):
            frame, ratios, best, n_init = self.pair.match_tile_auto_ksize(self.conf, invert_mon=inverted, candidates=LAPLACIAN_AUTO_CANDIDATES,
                                                                          origin=self.tile[:2])
            _log_search(ratios, best, inverted)
            return None if frame is None else _Trial(frame, n_init, best, inverted)
        # with the sigma clip the inlier count of a pair is only known after numpy's float32 statistics: one device run per pair
        winner, ratios = None, {}
        for mon_k in LAPLACIAN_AUTO_CANDIDATES:
            for ref_k in LAPLACIAN_AUTO_CANDIDATES:
                trial = self.fixed((mon_k, ref_k), inverted)
                ratios[(mon_k, ref_k)] = trial.inlier_ratio if trial else 0.0
                if trial and (winner is None or trial.inlier_ratio > winner.inlier_ratio):
                    winner = trial
        _log_search(ratios, winner.ksizes if winner else None, inverted)
        return winner

    def laplacians(self, trial: _Trial) -> tuple[np.ndarray, np.ndarray]:
        """(monitored, reference) Laplacian images of a trial, for the optional debug dump."""
        mon_box, ref_box = self._host
        lap_ref, lap_mon, _, _ = ops.tile_prefilter(ref_box, mon_box, ref_ksize=trial.ksizes[1], mon_ksize=trial.ksizes[0],
                                                    invert_mon=trial.inverted, with_mask=False, ctx=self.pair.ctx)
        return lap_mon, lap_ref


def _validity_mask(mon_box, ref_box, nodata_mon, nodata_ref) -> np.ndarray:
    """The automatic mask (klt.py:268-273) for pixel types the device does not read: 1 where both images hold a non-zero,
    finite value different from their no-data value."""
    valid = np.ones(mon_box.shape, bool)
    for box, nodata in ((mon_box, nodata_mon), (ref_box, nodata_ref)):
        valid &= box != 0
        if box.dtype.kind in "fc":
            valid &= np.isfinite(box)
        if nodata is not None:
            valid &= box != nodata
    return valid.view(np.uint8)


def _stretch_any(box: np.ndarray, ctx) -> np.ndarray:
    """Min-max stretch to uint8 with the reference's arithmetic (klt.py:42-49) for any numeric dtype: on the device for the
    types it reads, else numpy's own promotion rules."""
    if box.dtype == np.uint8:
        return box
    if box.dtype.type in _DEVICE_DTYPES:
        return ops.to_uint8(box, ctx=ctx)
    lo, hi = float(np.nanmin(box)), float(np.nanmax(box))
    if not hi > lo:
        return np.zeros(box.shape, np.uint8)
    return ((box - lo) / (hi - lo) * 255).astype(np.uint8)


def _log_search(ratios, best, inverted):
    if logger.isEnabledFor(logging.DEBUG):
        for pair, ratio in ratios.items():
            logger.debug("kernel-size search (%s polarity): mon %d / ref %d -> inlier ratio %.3f", "inverted" if inverted else "normal",
                         pair[0], pair[1], ratio)
    if best is None:
        logger.info("kernel-size search: no pair produced corners")
    else:
        logger.info("kernel-size search: mon %d / ref %d wins with inlier ratio %.3f", best[0], best[1], ratios[best])


class KLT:
    """Tile-wise KLT matching of a monitored image against a reference image."""

    def __init__(self, conf, gen_laplacian: bool = False, out_dir: str | None = None, ctx=None):
        self._conf = conf
        self._gen_laplacian = gen_laplacian
        self._out_dir = out_dir
        self._ctx = ctx
        self._ksize_votes: Counter = Counter()      # (mon, ref) chosen per tile by the kernel-size search
        self._polarity_votes: Counter = Counter()   # "normal" / "inverted" chosen per tile by the polarity search
        self._prefetched: dict = {}                 # identity of the rasters -> first tile session queued by `prefetch`

    # ------------------------------------------------------------------ public surface
    def match(self, mon_img, ref_img, mask) -> Iterator[DataFrame]:
        """Match every tile of the pair; yields the frames of the tiles that produced key points (klt.py:198-234)."""
        grid = self.tile_boxes(mon_img.x_size, mon_img.y_size)
        logger.info("KLT: %dx%d px in %d tile(s) of %d px, polarity %s, Laplacian kernel %s", mon_img.x_size, mon_img.y_size, len(grid),
                    self._conf.tile_size, self._describe_polarity(), self._conf.laplacian_kernel_size)
        if self._rasters_in_hbm(mon_img, ref_img, mask):
            yield from self._match_resident(grid, mon_img, ref_img, mask)
            return
        upcoming = self._prefetched.pop((id(mon_img), id(ref_img), id(mask)), None) if grid else None
        for k, tile in enumerate(grid):
            session = upcoming or self._open(tile, mon_img, ref_img, mask)
            # the next tile is read and its upload queued BEFORE this one is matched: from page-locked buffers the copy
            # travels while the device works on the current tile
            upcoming = self._open(grid[k + 1], mon_img, ref_img, mask) if k + 1 < len(grid) else None
            frame = self._finish(session)
            if frame is not None:
                yield frame
        if self._conf.laplacian_invert_polarity == "auto":
            total = sum(self._polarity_votes.values())
            if total:
                logger.info("polarity search: %s", ", ".join(f"{name} kept on {n}/{total} tiles" for name, n in self._polarity_votes.most_common()))
            else:
                logger.info("polarity search: no tile produced key points")

    def prefetch(self, mon_img, ref_img, mask=None) -> None:
        """Not in the reference: start reading and uploading the first tile of the NEXT pair now.  Called before the current
        pair's `match` is consumed, the copy (asynchronous from page-locked rasters, `karios_amd.pinned_empty`) travels while
        the device matches the current pair; the following `match(mon_img, ref_img, mask)` picks the tile up."""
        grid = self.tile_boxes(mon_img.x_size, mon_img.y_size)
        if grid:
            while len(self._prefetched) >= 2:           # a caller that prefetches without matching must not pile up tiles in HBM
                self._prefetched.pop(next(iter(self._prefetched)))
            self._prefetched[(id(mon_img), id(ref_img), id(mask))] = self._open(grid[0], mon_img, ref_img, mask)

    @property
    def auto_selected_ksize(self) -> tuple[int, int] | None:
        """The (mon, ref) kernel-size pair the search chose most often (None outside 'auto' mode / before `match`)."""
        return self._ksize_votes.most_common(1)[0][0] if self._ksize_votes else None

    @property
    def auto_selected_polarity(self) -> str | None:
        """'normal' or 'inverted', whichever the polarity search chose most often (None outside 'auto' mode)."""
        return self._polarity_votes.most_common(1)[0][0] if self._polarity_votes else None

    # ------------------------------------------------------------------ helpers other modules and tests rely on
    def tile_boxes(self, x_size: int, y_size: int) -> list[tiling.Tile]:
        return tiling.tile_grid(x_size, y_size, self._conf.tile_size, self._conf.xStart)

    _resolve_ksize = staticmethod(tiling.kernel_sizes)

    # ------------------------------------------------------------------ rasters that already live in HBM
    def _rasters_in_hbm(self, mon_img, ref_img, mask) -> bool:
        """Both rasters are `DeviceRasterImage`s of one pixel type the kernels read, the run uses fixed parameters and neither the
        sigma clip nor the Laplacian dump: the tiles are boxes of ONE resident pair - nothing is read, copied or uploaded."""
        from ..core.image import DeviceRasterImage
        conf = self._conf
        return (isinstance(mon_img, DeviceRasterImage) and isinstance(ref_img, DeviceRasterImage) and (mask is None or isinstance(mask, DeviceRasterImage))
                and mon_img.dtype == ref_img.dtype and mon_img.dtype.type in _DEVICE_DTYPES and conf.laplacian_kernel_size != "auto"
                and conf.laplacian_invert_polarity != "auto" and not getattr(conf, "outliers_filtering", False) and not self._gen_laplacian
                and mon_img.tensor.is_contiguous() and ref_img.tensor.is_contiguous() and (mask is None or mask.tensor.is_contiguous())
                and mon_img.tensor.shape == ref_img.tensor.shape)

    def _match_resident(self, grid, mon_img, ref_img, mask) -> Iterator[DataFrame]:
        """`match` on resident rasters: every tile of the grid is a box of the pair (pointer + stride), submitted through
        `karios_amd.stream.FrameStream` - tile i + 1 runs on the device while tile i becomes its DataFrame; frames in tile order."""
        import torch
        torch.cuda.synchronize(mon_img.tensor.device)               # the library works on its own streams
        pair = ResidentPair.from_device_pointers(mon_img.tensor.data_ptr(), ref_img.tensor.data_ptr(), mon_img.dtype, mon_img.y_size, mon_img.x_size,
                                                 ctx=self._ctx, mask_ptr=None if mask is None else mask.tensor.data_ptr(),
                                                 no_data_mon=getattr(mon_img, "no_data_value", None), no_data_ref=getattr(ref_img, "no_data_value", None),
                                                 keepalive=(mon_img.tensor, ref_img.tensor, None if mask is None else mask.tensor))
        for tile, frame in zip(grid, pair.match_pipelined(self._conf, boxes=[tuple(t) for t in grid], with_empty=True)):
            if frame is None:
                logger.info("tile (%d, %d): no valid pixels or no corners - skipped", tile[0], tile[1])
                continue
            n_init = int(frame.attrs.pop("Ninit", len(frame)))
            frame.attrs.clear()
            logger.info("tile (%d, %d): %d of %d corners kept", tile[0], tile[1], len(frame), n_init)
            yield frame

    # ------------------------------------------------------------------ one tile
    def _match_tile(self, x_off, y_off, mon_img, ref_img, mask) -> DataFrame | None:
        """Frame of the tile whose origin is (x_off, y_off), or None (no valid pixel, no corner)."""
        size = self._conf.tile_size
        tile = tiling.Tile(x_off, y_off, min(size, mon_img.x_size - x_off), min(size, mon_img.y_size - y_off))
        return self._finish(self._open(tile, mon_img, ref_img, mask))

    def _open(self, tile: tiling.Tile, mon_img, ref_img, mask) -> _TileSession:
        window = (1, *tile)
        return _TileSession(self._conf, tile, mon_img.read(*window), ref_img.read(*window), mask.read(*window) if mask else None,
                            getattr(mon_img, "no_data_value", None), getattr(ref_img, "no_data_value", None), self._ctx, rasters=(mon_img, ref_img))

    def _finish(self, session: _TileSession) -> DataFrame | None:
        x_off, y_off = session.tile[:2]
        trial = self._best_trial(session)
        if trial is None:
            logger.info("tile (%d, %d): no valid pixels or no corners - skipped", x_off, y_off)
            fixed_run = self._conf.laplacian_kernel_size != "auto" and self._conf.laplacian_invert_polarity != "auto"
            if self._gen_laplacian and fixed_run and session.valid_pixels > 0:   # the reference still writes its Laplacians then
                self._dump(session, _Trial(DataFrame(), 0, tiling.kernel_sizes(self._conf.laplacian_kernel_size),
                                           bool(self._conf.laplacian_invert_polarity)))
            return None
        if self._conf.laplacian_kernel_size == "auto":
            self._ksize_votes[trial.ksizes] += 1
        if self._gen_laplacian:
            self._dump(session, trial)
        pts = trial.frame
        pts.attrs.clear()
        logger.info("tile (%d, %d): %d of %d corners kept, mean shift (%.4f, %.4f), std (%.4f, %.4f)", x_off, y_off, len(pts), trial.n_init,
                    *(float(v) if len(pts) else float("nan") for v in (pts.dx.mean(), pts.dy.mean(), pts.dx.std(), pts.dy.std())))
        return pts

    def _best_trial(self, session: _TileSession) -> _Trial | None:
        search_kernels = self._conf.laplacian_kernel_size == "auto"
        run = session.kernel_search if search_kernels else (lambda inv: session.fixed(tiling.kernel_sizes(self._conf.laplacian_kernel_size), inv))
        mode = self._conf.laplacian_invert_polarity
        if mode != "auto":
            return run(bool(mode))
        # both polarities; the higher inlier ratio wins and 'normal' keeps a tie (klt.py:438-463)
        trials = [t for t in (run(False), run(True)) if t is not None]
        if not trials:
            return None
        best = max(trials, key=lambda t: t.inlier_ratio)   # max() returns the first of equal maxima: 'normal'
        self._polarity_votes[best.polarity] += 1
        logger.info("polarity search: %s kept (inlier ratio %.3f)", best.polarity, best.inlier_ratio)
        return best

    def _describe_polarity(self) -> str:
        mode = self._conf.laplacian_invert_polarity
        return "searched per tile" if mode == "auto" else ("inverted (255 - pixel of the monitored image)" if mode else "normal")

    def _dump(self, session: _TileSession, trial: _Trial) -> None:
        """Debug output of the two Laplacians of a tile, named like the reference's (klt.py:307-322)."""
        lap_mon, lap_ref = session.laplacians(trial)
        box = "_".join(str(v) for v in session.tile)
        stems = (f"mon_laplacian{'_inv' if trial.inverted else ''}_k{trial.ksizes[0]}_{box}", f"ref_laplacian_k{trial.ksizes[1]}_{box}")
        for stem, lap in zip(stems, (lap_mon, lap_ref)):
            path = os.path.join(self._out_dir or ".", stem)
            try:
                from skimage import io  # type: ignore
                io.imsave(path + ".tif", lap)
            except ImportError:
                np.save(path + ".npy", lap)
