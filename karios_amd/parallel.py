"""Tile-parallel matching over the GPUs of one node: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI;
"gloo" on CPU for tests), independent work units, ONE gather of the per-unit key-point blocks.

The reference's tiles are independent by construction - no halo, per-tile uint8 stretch, per-tile quality threshold and
maxCorners (`karios/matcher/klt.py:220-253`) - and so are the bands of a product, so unit *u* computed on any rank equals the
reference's result for that tile.  A rank uploads only the boxes of its own units (plus a margin of raw pixels for the ZNCC
chips, which the reference cuts from the full image, `zncc_service.py:289-297`); the device pipeline leaves every unit's
frame block (layout of km_klt_tile_frame_zncc_dev) directly in the rank's slice of the send buffer, and the only exchange
step is one flat all-gather of those fixed-size blocks (<= 640 KB per unit): latency-bound, never ring-chunked
(SURVEY.md 8e).  Frames are then assembled in unit order = the order `KLT.match` yields them, band after band.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
from pandas import DataFrame

from . import frames, tiling

COLUMNS = ["x0", "y0", "dx", "dy", "score"]
ZNCC_CHIP_MARGIN = 28          # (57 - 1) / 2: the reference's bounds rule is stated on the 57 x 57 chip
DEFAULT_HALO = 128             # raw pixels kept around a unit's box: ZNCC chips of displacements up to 99 px stay inside


@dataclass(frozen=True)
class WorkUnit:
    """One tile of one image pair ("band"), in the reference's enumeration order."""
    index: int      # global order: band major, then x_off outer / y_off inner (klt.py:220-232)
    band: int
    x_off: int
    y_off: int
    x_size: int
    y_size: int

    @property
    def box(self):
        return (self.x_off, self.y_off, self.x_size, self.y_size)


def enumerate_units(n_bands: int, x_size: int, y_size: int, conf) -> list[WorkUnit]:
    grid = tiling.tile_grid(x_size, y_size, conf.tile_size, conf.xStart)
    return [WorkUnit(b * len(grid) + k, b, *tile) for b in range(n_bands) for k, tile in enumerate(grid)]


def units_of_rank(units: list[WorkUnit], rank: int, world_size: int) -> list[WorkUnit]:
    """Static round-robin by unit index (equal-size units except edge tiles)."""
    return [u for u in units if u.index % world_size == rank]


def frame_capacity(conf, units) -> int:
    """Rows a unit's frame block must hold: maxCorners, or a quarter of the largest box when maxCorners is unbounded (0)."""
    if conf.maxCorners > 0:
        return int(conf.maxCorners)
    return max((max(1, (u.x_size * u.y_size) // 4) for u in units), default=1)


def block_len(cap: int, with_zncc: bool) -> int:
    """float32 words of one frame block (layout of km_klt_tile_frame[_zncc]_dev, see include/karios_hip.h)."""
    return 4 + (8 if with_zncc else 6) * cap


# ---------------------------------------------------------------------------- one unit on the device
class ResidentUnit:
    """A work unit's pixels in HBM: its box plus `halo` raw pixels on every side (clipped to the image)."""

    def __init__(self, unit: WorkUnit, pair, region, halo: int):
        self.unit, self.pair, self.region, self.halo = unit, pair, region, halo     # region = (x, y, w, h) in image coordinates

    @classmethod
    def load(cls, unit: WorkUnit, mon_img, ref_img, mask_img=None, halo: int = DEFAULT_HALO, ctx=None):
        """Read ONLY this unit's region through the raster accessors (`.read(1, x, y, w, h)`, `.x_size`, `.y_size`,
        `.no_data_value`: the GdalRasterImage duck type) and upload it."""
        from .resident import ResidentPair
        rx, ry = max(0, unit.x_off - halo), max(0, unit.y_off - halo)
        rw = min(mon_img.x_size, unit.x_off + unit.x_size + halo) - rx
        rh = min(mon_img.y_size, unit.y_off + unit.y_size + halo) - ry
        mask = None if mask_img is None else mask_img.read(1, rx, ry, rw, rh)
        mon_box, ref_box = mon_img.read(1, rx, ry, rw, rh), ref_img.read(1, rx, ry, rw, rh)
        from .core.image import DeviceWindow
        make = ResidentPair.from_windows if isinstance(mon_box, DeviceWindow) else ResidentPair.upload   # rasters already in HBM: device copies
        pair = make(mon_box, ref_box, mask, ctx=ctx, no_data_mon=getattr(mon_img, "no_data_value", None),
                    no_data_ref=getattr(ref_img, "no_data_value", None))
        # key points are image coordinates from the first kernel on: the float32 sum x0 + dx that decides which pixel a ZNCC
        # chip is centred on depends on the magnitude of x0 (zncc_service.py:195-196)
        pair.window = (rx, ry, mon_img.y_size, mon_img.x_size)
        return cls(unit, pair, (rx, ry, rw, rh), halo)

    @property
    def local_box(self):
        u, (rx, ry, _, _) = self.unit, self.region
        return (u.x_off - rx, u.y_off - ry, u.x_size, u.y_size)

    def match(self, conf, zncc_threshold=None) -> DataFrame | None:
        """The unit's frame in IMAGE coordinates (what `KLT._match_tile` returns for this tile, plus `zncc_score`)."""
        frame = self.pair.match_tile(conf, box=self.local_box, zncc_threshold=zncc_threshold, origin=(self.unit.x_off, self.unit.y_off))
        if frame is not None and zncc_threshold is not None:
            self.check_window(frame["zncc_score"].to_numpy())
        return frame

    def check_window(self, scores: np.ndarray) -> None:
        """A chip inside the image but outside the resident region cannot be scored here (the reference scores it): the
        device marks such rows (KM_NAN_OUTSIDE_WINDOW) instead of inventing a value."""
        from ._lib import NAN_OUTSIDE_WINDOW
        miss = int((np.ascontiguousarray(scores, np.float64).view(np.uint64) == NAN_OUTSIDE_WINDOW).sum())
        if miss:
            raise ValueError(f"unit {self.unit.index}: {miss} key point(s) moved further than the {self.halo} px halo covers "
                             f"({self.halo - ZNCC_CHIP_MARGIN} px); load the units with a larger halo")


# ---------------------------------------------------------------------------- the exchange step
def _world():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(), dist.get_rank()
    return 1, 0


def gather_block_tensor(send, n_units: int):
    """All-gather of every rank's `send` tensor, rows = (unit id | frame block), unit id < 0 for padding rows.  `send` lives
    where the collective runs (RCCL: HBM, gloo: host memory) and the result stays there.
    -> tensor (n_units, block_len) in unit order; units nobody computed have an all-zero block."""
    import torch
    import torch.distributed as dist
    ws, _ = _world()
    if ws == 1:
        recv = send
    else:
        recv = torch.empty((ws * send.shape[0], send.shape[1]), dtype=send.dtype, device=send.device)
        dist.all_gather_into_tensor(recv, send)
    # the blocks are moved as 32-bit integers: they carry int32 headers / labels and the halves of float64 scores, and a
    # float32 copy kernel may flush the words that happen to look like denormals
    out = torch.zeros((n_units, send.shape[1] - 1), dtype=torch.int32, device=send.device)
    ids = recv[:, 0].to(torch.int64)
    rows = torch.nonzero(ids >= 0).squeeze(1)
    out[ids[rows]] = recv.view(torch.int32)[rows, 1:]
    return out.view(torch.float32)


def gather_blocks(local: dict[int, np.ndarray | None], n_units: int, cap: int, with_zncc: bool = False, device=None) -> np.ndarray:
    """`gather_block_tensor` for blocks that sit in host memory (tests, CPU runs): `local` maps unit index -> block or None.
    Returns a (n_units, block_len) float32 array in unit order on every rank."""
    import torch
    import torch.distributed as dist
    L = block_len(cap, with_zncc)
    ws, _ = _world()
    per_rank = (n_units + ws - 1) // ws
    send = np.zeros((per_rank, 1 + L), np.float32)
    send[:, 0] = -1
    for slot, (idx, blk) in enumerate(sorted(local.items())):
        send[slot, 0] = idx
        if blk is not None:
            send[slot, 1:] = blk[:L]
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if ws > 1 and dist.get_backend() == "nccl" else torch.device("cpu")
    return gather_block_tensor(torch.from_numpy(send).to(device), n_units).cpu().numpy()


def gather_rank_blocks(block: np.ndarray | None, cap: int, with_zncc: bool = False, device=None):
    """One unit per rank (band-parallel run): all-gather of every rank's frame block, left on `device`.
    -> (tensor (world, block_len) float32 in rank order, total key points)."""
    import torch
    import torch.distributed as dist
    L = block_len(cap, with_zncc)
    ws, _ = _world()
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if ws > 1 and dist.get_backend() == "nccl" else torch.device("cpu")
    mine = torch.zeros(L, dtype=torch.float32) if block is None else torch.from_numpy(np.ascontiguousarray(block[:L]))
    t_send = mine.to(device)
    if ws == 1:
        t_recv = t_send.reshape(1, L)
    else:
        t_recv = torch.empty((ws, L), dtype=torch.float32, device=device)
        dist.all_gather_into_tensor(t_recv.reshape(-1), t_send)
    total = int(t_recv[:, 0].contiguous().view(torch.int32).sum().item())      # header word 0 = n_rows (int32 bit pattern)
    return t_recv, total


def blocks_to_frames(blocks: np.ndarray, cap: int, with_zncc: bool = False) -> list[DataFrame | None]:
    return [frames.block_to_frame(b, cap, with_zncc) for b in blocks]


# ---------------------------------------------------------------------------- frames as blocks (host side)
def pack_frame(frame: DataFrame | None, cap: int, with_zncc: bool = False) -> np.ndarray:
    """DataFrame (columns x0, y0, dx, dy, score [, zncc_score], any index) -> frame block, the inverse of
    `frames.block_to_frame`: the unit of the exchange step for frames that were finished on the host."""
    blk = np.zeros(block_len(cap, with_zncc), np.float32)
    if frame is None:
        return blk
    n = len(frame)
    if n > cap:
        raise ValueError(f"frame of {n} rows exceeds capacity {cap}")
    blk[:2] = np.array([n, int(frame.attrs.get("Ninit", max(n, 1)))], np.int32).view(np.float32)
    for k, name in enumerate(COLUMNS):
        blk[4 + k * cap:4 + k * cap + n] = frame[name].to_numpy(np.float32)
    blk[4 + 5 * cap:4 + 5 * cap + n] = frame.index.to_numpy().astype(np.int32).view(np.float32)
    if with_zncc:
        z = np.full(cap, np.nan)
        if "zncc_score" in frame.columns:
            z[:n] = frame["zncc_score"].to_numpy(np.float64)
        blk[4 + 6 * cap:] = z.view(np.float32)
    return blk


def unpack_frame(blk: np.ndarray, cap: int | None = None, with_zncc: bool = False) -> DataFrame | None:
    if cap is None:
        cap = (len(blk) - 4) // (8 if with_zncc else 6)
    return frames.block_to_frame(blk, cap, with_zncc)


def gather_frames(local: dict[int, DataFrame | None], n_units: int, cap: int, with_zncc: bool = False, device=None):
    """All-gather per-unit frames that exist as DataFrames; every rank returns them in unit order (None = no key points),
    rows, values, index labels and the optional zncc column intact."""
    blocks = gather_blocks({i: pack_frame(f, cap, with_zncc) for i, f in local.items()}, n_units, cap, with_zncc, device)
    return blocks_to_frames(blocks, cap, with_zncc)


# ---------------------------------------------------------------------------- the whole job
def match_distributed(bands: dict, n_bands: int, x_size: int, y_size: int, conf, score: bool = False,
                      confidence_threshold: float = 0.4, halo: int = DEFAULT_HALO, ctx=None, device=None):
    """Match `n_bands` image pairs tile-parallel over the ranks of the initialised process group.

    `bands[b] = (mon_img, ref_img[, mask_img])` raster accessors for every band of which this rank owns a unit
    (`units_of_rank`); only the units' regions are read and uploaded.  With `score` the frames carry `zncc_score`
    (rows with score >= confidence_threshold), `radial error` and `angle` like `_handle_klt_results` (core.py:872-893).
    Returns the frames in reference order on every rank."""
    import torch
    from ._lib import default_context
    ws, rank = _world()
    ctx = ctx if ctx is not None else default_context()
    units = enumerate_units(n_bands, x_size, y_size, conf)
    mine = units_of_rank(units, rank, ws)
    cap = frame_capacity(conf, units)
    L = block_len(cap, score)
    per_rank = (len(units) + ws - 1) // ws
    on_gpu = device is None or torch.device(device).type == "cuda"
    dev = torch.device("cuda", ctx.device) if on_gpu else torch.device(device)
    send = torch.zeros((per_rank, 1 + L), dtype=torch.float32, device=dev)
    send[:, 0] = -1
    thr = confidence_threshold if score else None
    for slot, u in enumerate(mine):
        src = bands[u.band]
        ru = ResidentUnit.load(u, src[0], src[1], src[2] if len(src) > 2 else None, halo=halo, ctx=ctx)
        if on_gpu and not getattr(conf, "outliers_filtering", False):
            # the device pipeline drops the unit's block straight into the send buffer (km_set_frame_sink)
            ctx.check(ctx.lib.km_set_frame_sink(ctx.handle, send[slot, 1:].data_ptr(), L * 4), "km_set_frame_sink")
            try:
                ru.match(conf, thr)
            finally:
                ctx.check(ctx.lib.km_set_frame_sink(ctx.handle, None, 0), "km_set_frame_sink")
        else:
            frame = ru.match(conf, thr)
            send[slot, 1:] = torch.from_numpy(pack_frame(frame, cap, score)).to(dev)
        send[slot, 0] = u.index
    if on_gpu:
        ctx.sync()                      # the library's stream wrote the sinks; the collective runs on torch's
        torch.cuda.synchronize(dev)
    blocks = gather_block_tensor(send, len(units)).cpu().numpy()
    out = blocks_to_frames(blocks, cap, score)
    if score:
        out = [None if f is None else frames.radial_angle_columns(f) for f in out]
    return out
