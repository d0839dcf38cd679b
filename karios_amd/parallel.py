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


def block_len(cap: int, with_zncc=False) -> int:
    """float32 words of one frame block (layout of km_klt_tile_frame[_zncc]_dev, see include/karios_hip.h; `with_zncc` = number
    of float64 score columns: False / True / 3, `frames.block_words`)."""
    return frames.block_words(cap, with_zncc)


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


def _group_up() -> bool:
    """A process group exists: the collectives are ISSUED even when it has a single rank (the RCCL / gloo code path of a one-rank
    job is the one an eight-rank job takes; without a group the exchange steps reduce to identities)."""
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized()


def _collective_device(ctx=None, device=None):
    """Where the collectives' tensors live: HBM for RCCL ("nccl"), host memory for gloo / no group."""
    import torch
    import torch.distributed as dist
    if device is not None:
        return torch.device(device)
    if _group_up() and dist.get_backend() == "nccl":
        return torch.device("cuda", ctx.device if ctx is not None else torch.cuda.current_device())
    return torch.device("cpu")


def gather_block_tensor(send, n_units: int):
    """All-gather of every rank's `send` tensor, rows = (unit id | frame block), unit id < 0 for padding rows.  `send` lives
    where the collective runs (RCCL: HBM, gloo: host memory) and the result stays there.
    -> tensor (n_units, block_len) in unit order; units nobody computed have an all-zero block."""
    import torch
    import torch.distributed as dist
    ws, _ = _world()
    if not _group_up():
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
    device = _collective_device(device=device)
    return gather_block_tensor(torch.from_numpy(send).to(device), n_units).cpu().numpy()


def gather_rank_blocks(block: np.ndarray | None, cap: int, with_zncc: bool = False, device=None):
    """One unit per rank (band-parallel run): all-gather of every rank's frame block, left on `device`.
    -> (tensor (world, block_len) float32 in rank order, total key points)."""
    import torch
    import torch.distributed as dist
    L = block_len(cap, with_zncc)
    ws, _ = _world()
    device = _collective_device(device=device)
    mine = torch.zeros(L, dtype=torch.float32) if block is None else torch.from_numpy(np.ascontiguousarray(block[:L]))
    t_send = mine.to(device)
    if not _group_up():
        t_recv = t_send.reshape(1, L)
    else:
        t_recv = torch.empty((ws, L), dtype=torch.float32, device=device)
        dist.all_gather_into_tensor(t_recv.reshape(-1), t_send)
    total = int(t_recv[:, 0].contiguous().view(torch.int32).sum().item())      # header word 0 = n_rows (int32 bit pattern)
    return t_recv, total


class RankBlockExchange:
    """The exchange step of a band-parallel STREAM of units (one unit per rank and step, `klt.py:220-253`: tiles / bands are
    independent; SURVEY 8e: one flat all-gather of the fixed-size key-point blocks) without the host in the loop.

    RCCL: the library writes step s's frame block straight into a slot of a send ring in HBM (`km_set_frame_sink`); when `batch` steps
    have been collected, a side stream waits ON THE DEVICE for their blocks (`km_stream_wait_frame`: satisfied at once, the steps are
    finished) and ONE asynchronous all-gather moves the `batch` blocks of every rank; one small device copy keeps the gathered
    blocks' 16-byte headers, from which rows / flagged blocks are summed behind the last step - nothing is read back before
    `finish()`.  The ring has two halves of `batch` slots: one fills while the other travels.  gloo (development: ranks sharing a
    GPU): the host block of every step is gathered asynchronously and waited for a few steps later.

        ex = RankBlockExchange(ctx, cap, n_scores)
        for s in range(steps):
            ex.arm(s)                                            # frame sink -> send slot of step s
            for d in stream.submit(pair, conf, tag=s, on_submitted=lambda p, s=s: pend.__setitem__(s, p)):
                ex.issue(d.tag, pend.pop(d.tag))                 # a COLLECTED step, in submission order
        ...                                                      # (same for stream.drain())
        rows, flagged = ex.finish()

    Why the exchange of a step is issued when the step is COLLECTED and not when it is submitted: a device-side wait issued at
    submission sits in a hardware queue for the whole step, and the runtime maps more streams than it has hardware queues onto
    shared queues - whenever the side stream shared one with the library's streams the next unit stalled behind it (measured: 6 %
    against 3 %).  Why `batch`: a collective costs the device ~3 % of a 1.1-ms step (its workgroups take slots from the dense kernels
    of the units behind); four steps per collective leave < 1 %.
    """
    HISTORY = 4096                    # gathered headers kept on the device before they are folded into the counters (in steps)

    def __init__(self, ctx, cap: int, with_zncc=True, device=None, batch: int = 4, slots: int | None = None, parts: str = "sink,wait,gather,account",
                 halves: int = 2):
        import torch
        from collections import deque
        self.ctx, self.cap, self.L = ctx, cap, block_len(cap, with_zncc)
        self.ws, self.rank = _world()
        self.grouped = _group_up()
        self.device = _collective_device(ctx, device)
        self.on_gpu = self.device.type == "cuda"
        self.batch = max(1, int(batch)) if self.on_gpu else 1
        self.halves = max(2, int(halves))     # groups of `batch` send slots (RCCL): with batched submissions of `batch` steps and a stream depth d, d + 1 groups
        self.slots = self.halves * self.batch if self.on_gpu else max(2, int(slots or 4))
        self.parts = set(parts.split(","))          # (tools/exchange_probe.py switches stages off to price them; everything on otherwise)
        halves = self.halves if self.on_gpu else self.slots
        per = self.batch if self.on_gpu else 1
        self.send = torch.zeros((halves, per, self.L), dtype=torch.float32, device=self.device)
        self.recv = torch.zeros((halves, self.ws, per, self.L), dtype=torch.float32, device=self.device)
        self.hist = torch.zeros((self.HISTORY, self.ws, 4), dtype=torch.int32, device=self.device)
        self.hist_n = 0
        self.rows = torch.zeros((), dtype=torch.int64, device=self.device)
        self.flagged = torch.zeros((), dtype=torch.int64, device=self.device)
        self.side = torch.cuda.Stream(self.device) if self.on_gpu else None
        self.done = [torch.cuda.Event() if self.on_gpu else None for _ in range(halves)]   # the half's collective + header copy have finished
        self.used = [False] * halves
        self.filled = [0] * halves                 # slot index behind the last step collected into the half since its last collective
        self.first = [0] * halves                  # slot index of the first one (a run may start or restart in the middle of a half)
        self.pending = deque()                     # gloo: (work, slot) issued and not waited for yet
        self.issued = 0
        self._sink_set = False
        self.slot_step = [[None] * per for _ in range(halves)]    # RCCL: the step whose block a send slot holds until its batch has been gathered

    def _where(self, step: int):
        if self.on_gpu:
            return (step // self.batch) % self.halves, step % self.batch
        return step % self.slots, 0

    def arm(self, step: int) -> None:
        """Before submitting step `step`: its block goes to the step's send slot (RCCL).  The half's previous collective is `batch`
        steps old at least; the check is free.  A slot whose previous block has not been gathered yet is never re-armed: a half is
        gathered when its last step is COLLECTED, i.e. `FrameStream.depth` submissions later, so the ring needs depth <= batch."""
        if not self.on_gpu or "sink" not in self.parts:
            return
        half, j = self._where(step)
        if "wait" in self.parts:
            old = self.slot_step[half][j]
            if old is not None:
                raise RuntimeError(f"RankBlockExchange: step {step} would overwrite the block of step {old}, which has not been gathered yet "
                                   f"(steps are gathered in batches of {self.batch} when they are collected: keep the stream's depth <= batch)")
            self.slot_step[half][j] = step
        if j == 0 and self.used[half]:
            self.done[half].synchronize()
        c = self.ctx
        c.set_frame_sink(self.send[half, j].data_ptr(), self.L * 4)
        self._sink_set = True

    def arm_many(self, first_step: int, n: int) -> None:
        """Before a BATCHED submission of the steps `first_step .. first_step + n - 1` (`FrameStream.submit_many`: one device pipeline
        for n units): their blocks go to n consecutive send slots of ONE half (pitch = the block size), so `(first_step % batch) + n
        <= batch`."""
        if not self.on_gpu or "sink" not in self.parts:
            return
        half, j = self._where(first_step)
        if j + n > self.batch:
            raise ValueError(f"RankBlockExchange.arm_many: steps {first_step} .. {first_step + n - 1} straddle a half of {self.batch} slots")
        if "wait" in self.parts:
            for k in range(n):
                old = self.slot_step[half][j + k]
                if old is not None:
                    raise RuntimeError(f"RankBlockExchange: step {first_step + k} would overwrite the block of step {old}, which has not been gathered yet "
                                       f"(keep the stream's depth x units per submission <= the ring's {self.halves * self.batch} slots)")
                self.slot_step[half][j + k] = first_step + k
        if j == 0 and self.used[half]:
            self.done[half].synchronize()
        self.ctx.set_frame_sink(self.send[half, j].data_ptr(), n * self.L * 4, self.L * 4)
        self._sink_set = True

    def issue_many(self, first_step: int, n: int, pending) -> None:
        """The collected steps `first_step .. first_step + n - 1` of ONE batched submission (`pending`: its PendingBatch): one device-side
        wait covers all n blocks."""
        if not self.on_gpu:
            raise RuntimeError("issue_many: RCCL path only (host blocks go through issue(step, host_block=...) one by one)")
        half, j = self._where(first_step)
        c = self.ctx
        if "wait" in self.parts and "sink" in self.parts:
            c.check(c.lib.km_stream_wait_frame(c.handle, pending.ticket, self.side.cuda_stream), "km_stream_wait_frame")
            if self.filled[half] == 0:
                self.first[half] = j
            self.filled[half] = j + n
            if j + n == self.batch:
                self._gather_half(half)
        self.issued += n

    def _fold(self) -> None:
        """History of headers -> counters (a handful of small device ops, once per HISTORY steps and at the end)."""
        import torch
        if self.hist_n == 0:
            return
        h = self.hist[:self.hist_n].reshape(-1, 4)
        good = h[:, 2] == 0
        self.rows += torch.where(good, h[:, 0], torch.zeros_like(h[:, 0])).sum()
        self.flagged += (~good).sum()
        self.hist_n = 0

    def _keep_headers(self, half: int, n_steps: int) -> None:
        if self.hist_n + n_steps > self.HISTORY:
            self._fold()
        # (ws, n_steps, 4) float32 -> int32 words, step-major in the history
        blockh = self.recv[half][:, :n_steps, :4].contiguous().view(self.hist.dtype)
        self.hist[self.hist_n:self.hist_n + n_steps].copy_(blockh.transpose(0, 1), non_blocking=True)
        self.hist_n += n_steps

    def _gather_half(self, half: int) -> None:
        """One collective for the blocks collected into `half`: slots [first, filled).  The other slots of the half (a run that starts,
        restarts or ends in the middle of a half) hold older blocks: they travel with zeroed headers and count nothing."""
        import torch
        import torch.distributed as dist
        lo, hi = self.first[half], self.filled[half]
        prev = torch.cuda.current_stream(self.device)
        torch.cuda.set_stream(self.side)
        try:
            if lo > 0:
                self.send[half, :lo, :4].zero_()
            if hi < self.batch:
                self.send[half, hi:, :4].zero_()
            if "gather" in self.parts:
                if self.grouped:
                    work = dist.all_gather_into_tensor(self.recv[half].view(-1), self.send[half].view(-1), async_op=True)
                    work.wait()                    # (orders the side stream behind the collective: no host wait)
                else:
                    self.recv[half][0].copy_(self.send[half], non_blocking=True)
            if "account" in self.parts:
                self._keep_headers(half, self.batch)
            self.done[half].record(self.side)
            self.used[half] = True
        finally:
            torch.cuda.set_stream(prev)
        for j in range(lo, hi):
            self.slot_step[half][j] = None
        self.filled[half] = 0

    def issue(self, step: int, pending=None, host_block=None) -> None:
        """A collected step (RCCL: `pending` = its PendingFrame, steps in submission order; gloo: `host_block` = its finished block)."""
        import torch
        import torch.distributed as dist
        half, j = self._where(step)
        if self.on_gpu:
            c = self.ctx
            if "wait" in self.parts and "sink" in self.parts:
                c.check(c.lib.km_stream_wait_frame(c.handle, pending.ticket, self.side.cuda_stream), "km_stream_wait_frame")
                if self.filled[half] == 0:
                    self.first[half] = j
                self.filled[half] = j + 1
                if j + 1 == self.batch:
                    self._gather_half(half)
        else:
            blk = np.zeros(self.L, np.float32) if host_block is None else np.ascontiguousarray(host_block[:self.L])
            self.send[half, 0].copy_(torch.from_numpy(blk))
            if self.grouped:
                self.pending.append((dist.all_gather_into_tensor(self.recv[half].view(-1), self.send[half].view(-1), async_op=True), half))
            else:
                self.recv[half][0].copy_(self.send[half])
                self.pending.append((None, half))
            while len(self.pending) > self.slots - 1:
                self._retire()
        self.issued += 1

    def _retire(self) -> None:
        work, half = self.pending.popleft()
        if work is not None:
            work.wait()
        self._keep_headers(half, 1)

    def finish(self) -> tuple[int, int]:
        """-> (matched key points of every rank and step, gathered blocks that were flagged by the synchronisation-free corner path
        - their rows are NOT counted: the owner repeated those units exactly and knows their rows).  Every rank must call it after the
        same number of steps (a partial batch is gathered here)."""
        import torch
        while self.pending:
            self._retire()
        if self.on_gpu:
            if self._sink_set:
                self.ctx.set_frame_sink(None)
                self._sink_set = False
            for half in range(self.halves):
                if self.filled[half]:
                    self._gather_half(half)
            with torch.cuda.stream(self.side):
                self._fold()
                both = torch.stack([self.rows, self.flagged])
            self.side.synchronize()
        else:
            self._fold()
            both = torch.stack([self.rows, self.flagged])
        rows, flagged = both.tolist()                 # ONE read-back
        return int(rows), int(flagged)

    def reset_counts(self) -> None:
        """Counters back to zero (call with everything issued so far finished: after `finish()`)."""
        self.rows.zero_(); self.flagged.zero_(); self.hist_n = 0

    def last_blocks(self, step: int):
        """The gathered blocks of step `step` (tensor (world, block_len), rank order) once its batch has travelled (`finish()` sends a
        partial one) and until the half is reused."""
        half, j = self._where(step)
        return self.recv[half][:, j]


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
        cap = (len(blk) - 4) // (6 + 2 * int(with_zncc))
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
            ctx.set_frame_sink(send[slot, 1:].data_ptr(), L * 4)
            try:
                ru.match(conf, thr)
            finally:
                ctx.set_frame_sink(None)
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


# ---------------------------------------------------------------------------- one tile, several GPUs, exactly
def band_rows(y_size: int, world_size: int) -> list[int]:
    """Row boundaries of the bands: world_size + 1 even numbers (the pyramids of a band must align with the image's)."""
    edges = [((y_size * r) // world_size) & ~1 for r in range(world_size)] + [y_size]
    if any(b <= a for a, b in zip(edges, edges[1:])):
        raise ValueError(f"{y_size} rows cannot be split into {world_size} bands")
    return edges


def match_tile_banded(mon_img, ref_img, mask_img, conf, zncc_threshold=None, halo: int = 96, ctx=None, device=None):
    """The reference's DEFAULT configuration - one tile covering the whole image (tile_size 20000, klt.py:220-253) - matched by
    all ranks of the process group together, with the single-GPU result: every rank reads and processes only the rows of its
    band plus `halo` rows, and between the device stages the ranks exchange

      * min / max of the two rasters (two all-reduces): the tile-wide uint8 stretch (klt.py:42-49);
      * the maximum eigenvalue (one all-reduce): goodFeaturesToTrack's quality threshold;
      * their strongest candidate keys (one all-gather, ~8 * maxCorners keys each): every rank then runs the SAME ranked
        greedy selection with the one maxCorners cut (klt.py:120) on the merged list;
      * the tracks of the corners they own (one all-reduce): the frame is assembled identically everywhere.

    Returns the tile's frame (columns x0, y0, dx, dy, score [, zncc_score], rows by (x0, y0)) or None, on every rank."""
    import ctypes as C
    import torch
    import torch.distributed as dist
    from ._lib import NAN_OUTSIDE_WINDOW, KariosHipError, as_image, default_context, dtype_code
    from .ops import make_params
    from .resident import DeviceBuffer, ResidentPair
    ws, rank = _world()
    ctx = ctx if ctx is not None else default_context()
    H, W = mon_img.y_size, mon_img.x_size
    if conf.tile_size < max(H, W) or conf.xStart > 0:
        raise ValueError("match_tile_banded reproduces the single-tile configuration (tile_size >= image size, xStart 0)")
    if conf.laplacian_kernel_size == "auto" or conf.laplacian_invert_polarity == "auto" or getattr(conf, "outliers_filtering", False):
        raise ValueError("match_tile_banded: fixed kernel sizes / polarity, no outlier filtering")
    if halo % 2 or halo < 32:
        raise ValueError("halo must be an even number of rows >= 32")
    coll = _collective_device(ctx, device)
    grouped = _group_up()

    def reduce_(values, op, dtype=torch.float64):
        t = torch.as_tensor(np.asarray(values), dtype=dtype).to(coll)
        if grouped:
            dist.all_reduce(t, op=op)
        return t.cpu().numpy()

    edges = band_rows(H, ws)
    y0, y1 = edges[rank], edges[rank + 1]
    ys, ye = max(0, y0 - halo), min(H, y1 + halo)
    hs = ye - ys
    mon = as_image(mon_img.read(1, 0, ys, W, hs))
    ref = as_image(ref_img.read(1, 0, ys, W, hs))
    if mon.dtype != ref.dtype:
        raise KariosHipError("match_tile_banded: both rasters must have the same pixel type")
    code = dtype_code(mon)
    pair = ResidentPair.upload(mon, ref, None if mask_img is None else mask_img.read(1, 0, ys, W, hs), ctx=ctx,
                               no_data_mon=getattr(mon_img, "no_data_value", None), no_data_ref=getattr(ref_img, "no_data_value", None))
    pair._ready()
    pair.window = (0, ys, H, W)
    lib, h = ctx.lib, ctx.handle
    # ---- tile-wide min / max
    mm = np.zeros(4)
    for k, ptr in ((0, pair.ref_ptr), (2, pair.mon_ptr)):
        out = (C.c_double * 2)()
        ctx.check(lib.km_minmax_dev(h, C.c_void_p(ptr), code, hs, W, W, out), "km_minmax_dev")
        mm[k], mm[k + 1] = out[0], out[1]
    if code != 0:     # (uint8 rasters pass through the stretch unchanged)
        lo = reduce_([mm[0], mm[2]], dist.ReduceOp.MIN)
        hi = reduce_([mm[1], mm[3]], dist.ReduceOp.MAX)
        mm = np.array([lo[0], hi[0], lo[1], hi[1]])
    # ---- stretch, Laplacians, mask of the band's own rows
    prm = make_params(conf, *tiling.kernel_sizes(conf.laplacian_kernel_size), bool(conf.laplacian_invert_polarity))
    lap_ref, lap_mon, mask = (DeviceBuffer(ctx, hs * W) for _ in range(3))
    valid = C.c_int64()
    nr = C.byref(C.c_double(float(pair.no_data_ref))) if pair.no_data_ref is not None else None
    nm = C.byref(C.c_double(float(pair.no_data_mon))) if pair.no_data_mon is not None else None
    ctx.check(lib.km_band_prefilter_dev(h, C.c_void_p(pair.ref_ptr), C.c_void_p(pair.mon_ptr), code, hs, W, W, W, mm.ctypes.data_as(C.POINTER(C.c_double)),
                                        nr, nm, prm.ksize_ref, prm.ksize_mon, prm.invert_mon, y0 - ys, y1 - ys,
                                        C.c_void_p(pair.mask_ptr) if pair.mask_ptr else None, C.c_void_p(lap_ref.ptr), C.c_void_p(lap_mon.ptr),
                                        C.c_void_p(mask.ptr), C.byref(valid)), "km_band_prefilter_dev")
    if int(reduce_([valid.value], dist.ReduceOp.SUM)[0]) == 0:
        return None                                                     # "No valid pixels" (klt.py:276-279)
    # ---- maximum eigenvalue of the tile
    key = C.c_uint()
    ctx.check(lib.km_band_eigen_dev(h, C.c_void_p(lap_ref.ptr), C.c_void_p(mask.ptr), hs, W, prm.block_size, prm.quality_level, C.byref(key)),
              "km_band_eigen_dev")
    gkey = int(reduce_([key.value], dist.ReduceOp.MAX, torch.int64)[0])
    if gkey == 0:
        return None
    # ---- candidates: every rank contributes its strongest value bins; the merged list is cut where the weakest contribution ends
    max_corners = int(conf.maxCorners)
    sliced = max_corners > 0 and conf.minDistance >= 1
    cap = max_corners if max_corners > 0 else max(1, (H * W) // 4)
    corners = None
    for k_target in ((8 * max_corners, 0) if sliced else (0,)):
        room = hs * W // 4 + 65536
        buf = np.empty(room, np.uint64)
        n_out, n_tot = C.c_size_t(), C.c_size_t()
        ctx.check(lib.km_band_keys_dev(h, C.c_void_p(mask.ptr), hs, W, prm.quality_level, gkey, k_target, buf.ctypes.data_as(C.c_void_p), room,
                                       C.byref(n_out), C.byref(n_tot)), "km_band_keys_dev")
        mine = buf[:n_out.value] + np.uint64(ys * W)                      # raster index of the band image -> of the tile
        truncated = n_out.value < n_tot.value
        floor_key = int(mine.min()) if truncated and len(mine) else 0     # nothing weaker than this is known from this rank
        counts = reduce_([len(mine) if r == rank else 0 for r in range(ws)], dist.ReduceOp.SUM, torch.int64)
        floors = reduce_(np.array([floor_key if r == rank else 0 for r in range(ws)], np.uint64).view(np.int64), dist.ReduceOp.SUM, torch.int64)
        width = max(1, int(counts.max()))
        send = torch.zeros(width, dtype=torch.int64)
        send[:len(mine)] = torch.from_numpy(mine.view(np.int64).copy())
        if grouped:
            flat = torch.empty(ws * width, dtype=torch.int64, device=coll)
            dist.all_gather_into_tensor(flat, send.to(coll))
            recv = flat.view(ws, width)
            parts = [recv[r, :int(counts[r])].cpu().numpy().view(np.uint64) for r in range(ws)]
        else:
            parts = [mine]
        merged = np.ascontiguousarray(np.concatenate(parts))
        cut = floors.view(np.uint64).max()
        merged = np.ascontiguousarray(merged[merged >= cut])               # a rank prefix of the tile's candidate list
        out_xy = np.empty((cap, 2), np.float32)
        n = C.c_int()
        ctx.check(lib.km_select_keys(h, merged.ctypes.data_as(C.c_void_p), len(merged), H, W, max_corners, float(conf.minDistance),
                                     out_xy.ctypes.data_as(C.c_void_p), cap, C.byref(n)), "km_select_keys")
        corners = out_xy[:n.value].copy()
        if not (sliced and k_target and n.value < max_corners and int(cut) > 0):
            break                                                          # enough corners, or the list was complete
    if corners is None or len(corners) == 0:
        return None                                                        # "No features extracted" (klt.py:122-124)
    # ---- tracks of the corners this rank owns
    own = (corners[:, 1] >= y0) & (corners[:, 1] < y1)
    p0_own = np.ascontiguousarray(corners[own])
    p1_own, p0r_own = np.empty_like(p0_own), np.empty_like(p0_own)
    left = C.c_int()
    ctx.check(lib.km_band_track_dev(h, C.c_void_p(lap_ref.ptr), C.c_void_p(lap_mon.ptr), hs, W, ys, H, C.byref(prm), p0_own.ctypes.data_as(C.c_void_p),
                                    len(p0_own), p1_own.ctypes.data_as(C.c_void_p), p0r_own.ctypes.data_as(C.c_void_p), C.byref(left)),
              "km_band_track_dev")
    if int(reduce_([left.value], dist.ReduceOp.MAX, torch.int64)[0]):
        raise KariosHipError(f"match_tile_banded: a tracked window left its band's {halo}-row halo; use a larger halo")
    tracks = np.zeros((2, len(corners), 2), np.float32)
    tracks[0, own], tracks[1, own] = p1_own, p0r_own
    tracks = reduce_(tracks, dist.ReduceOp.SUM, torch.float32)              # every corner has exactly one owner: the sums are copies
    cols, n_init = frames.track_columns(corners.reshape(-1, 1, 2), tracks[0].reshape(-1, 1, 2), tracks[1].reshape(-1, 1, 2))
    frame = frames.assemble(cols)
    if zncc_threshold is not None:
        z = np.zeros(len(frame))
        score_it = frame["score"].to_numpy() >= zncc_threshold
        mine_rows = score_it & (frame["y0"].to_numpy() >= y0) & (frame["y0"].to_numpy() < y1)
        outside = 0
        if mine_rows.any():
            sub = frame[mine_rows]
            vals = pair.zncc(sub["x0"].to_numpy(), sub["y0"].to_numpy(), sub["dx"].to_numpy(), sub["dy"].to_numpy())
            outside = int((np.ascontiguousarray(vals).view(np.uint64) == NAN_OUTSIDE_WINDOW).any())
            z[mine_rows] = np.where(np.isnan(vals), np.inf, vals)          # (a NaN would poison the sum: it travels as inf)
        if int(reduce_([outside], dist.ReduceOp.MAX, torch.int64)[0]):     # (decided together: nobody is left waiting in the sum)
            raise KariosHipError(f"match_tile_banded: a ZNCC chip left its band's {halo}-row halo; use a larger halo")
        z = reduce_(z, dist.ReduceOp.SUM)
        z[np.isinf(z)] = np.nan
        z[~score_it] = np.nan
        frame["zncc_score"] = z
    frame.attrs["Ninit"] = n_init
    return frame
