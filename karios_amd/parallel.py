"""Tile-parallel matching over the GPUs of one node: one process per GPU (torch.distributed,
backend "nccl" = RCCL over xGMI; "gloo" on CPU for tests), independent work units, one
gather of the per-unit key-point frames.

The reference's tiles are independent by construction -- no halo, per-tile uint8 stretch,
per-tile quality threshold and maxCorners (`karios/matcher/klt.py:220-253`) -- so unit *u*
computed on any rank equals the reference's result for that tile.  The only exchange step is
the final gather: fixed-size padded float32 blocks (<= 5*maxCorners floats + a count per
unit), a single flat all-gather -- latency-bound, never ring-chunked.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
from pandas import DataFrame

COLUMNS = ["x0", "y0", "dx", "dy", "score"]


@dataclass(frozen=True)
class WorkUnit:
    """One tile of one image pair ("band"), in the reference's enumeration order."""
    index: int      # global order: band major, then x_off outer / y_off inner (klt.py:220-232)
    band: int
    x_off: int
    y_off: int
    x_size: int
    y_size: int


def enumerate_units(n_bands: int, x_size: int, y_size: int, conf) -> list[WorkUnit]:
    from .matcher.klt import KLT
    units, i = [], 0
    for b in range(n_bands):
        for x_off, y_off, bx, by in KLT(conf).tile_boxes(x_size, y_size):
            units.append(WorkUnit(i, b, x_off, y_off, bx, by))
            i += 1
    return units


def units_of_rank(units: list[WorkUnit], rank: int, world_size: int) -> list[WorkUnit]:
    """Static round-robin by unit index (equal-size units except edge tiles)."""
    return [u for u in units if u.index % world_size == rank]


def pack_frame(frame: DataFrame | None, cap: int) -> np.ndarray:
    """DataFrame -> fixed-size float32 block [count, 5*cap values] (row-major rows of COLUMNS)."""
    blk = np.zeros(1 + 5 * cap, np.float32)
    if frame is not None and len(frame):
        n = len(frame)
        if n > cap:
            raise ValueError(f"frame of {n} rows exceeds capacity {cap}")
        blk[0] = n
        blk[1:1 + 5 * n] = frame[COLUMNS].to_numpy(np.float32).ravel()
    return blk


def unpack_frame(blk: np.ndarray) -> DataFrame | None:
    n = int(blk[0])
    if n == 0:
        return None
    return DataFrame(blk[1:1 + 5 * n].reshape(n, 5).copy(), columns=COLUMNS)


def gather_frames(local: dict[int, DataFrame | None], n_units: int, cap: int, device=None):
    """All-gather the per-unit frames of every rank; every rank returns the list of frames ordered by
    unit index (None for units without points) -- i.e. the order `KLT.match` yields them.

    `local` maps unit index -> frame for the units this rank computed.  Works on any initialised
    torch.distributed backend; with NCCL/RCCL the blocks travel GPU to GPU over xGMI."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [local.get(i) for i in range(n_units)]
    ws, rank = dist.get_world_size(), dist.get_rank()
    per_rank = (n_units + ws - 1) // ws
    blk_len = 1 + 5 * cap
    send = np.zeros((per_rank, 1 + blk_len), np.float32)
    send[:, 0] = -1  # unit id, -1 = padding slot
    for slot, (idx, frame) in enumerate(sorted(local.items())):
        send[slot, 0] = idx
        send[slot, 1:] = pack_frame(frame, cap)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    t_send = torch.from_numpy(send).to(device)
    # output = rank blocks concatenated along dim 0 (the layout both RCCL and gloo accept)
    t_recv = torch.empty((ws * per_rank, send.shape[1]), dtype=t_send.dtype, device=device)
    dist.all_gather_into_tensor(t_recv, t_send)
    recv = t_recv.cpu().numpy().reshape(ws, per_rank, send.shape[1])
    frames: list[DataFrame | None] = [None] * n_units
    for r in range(ws):
        for slot in range(per_rank):
            idx = int(recv[r, slot, 0])
            if idx >= 0:
                frames[idx] = unpack_frame(recv[r, slot, 1:])
    return frames


def block_len(cap: int, with_zncc: bool) -> int:
    """float32 words of one frame block (layout of km_klt_tile_frame[_zncc]_dev, see include/karios_hip.h)."""
    return 4 + (8 if with_zncc else 6) * cap


def gather_blocks(local: dict[int, np.ndarray | None], n_units: int, cap: int, with_zncc: bool = False, device=None) -> np.ndarray:
    """All-gather raw frame blocks (the device pipeline's own output layout) -- the per-step exchange of the tile-parallel
    run.  `local`: unit index -> block (or None).  Returns a (n_units, block_len) float32 array in unit order on every
    rank; a unit without result has an all-zero header.  One flat all-gather, sized for latency, not bandwidth."""
    import torch
    import torch.distributed as dist

    L = block_len(cap, with_zncc)
    out = np.zeros((n_units, L), np.float32)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        for idx, blk in local.items():
            if blk is not None:
                out[idx] = blk[:L]
        return out
    ws = dist.get_world_size()
    per_rank = (n_units + ws - 1) // ws
    send = np.zeros((per_rank, 1 + L), np.float32)
    send[:, 0] = -1
    for slot, (idx, blk) in enumerate(sorted(local.items())):
        send[slot, 0] = idx
        if blk is not None:
            send[slot, 1:] = blk[:L]
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    t_send = torch.from_numpy(send).to(device)
    t_recv = torch.empty((ws * per_rank, 1 + L), dtype=t_send.dtype, device=device)
    dist.all_gather_into_tensor(t_recv, t_send)
    recv = t_recv.cpu().numpy()
    ids = recv[:, 0].astype(np.int64)
    ok = ids >= 0
    out[ids[ok]] = recv[ok, 1:]
    return out


def gather_rank_blocks(block: np.ndarray | None, cap: int, with_zncc: bool = False, device=None):
    """The per-step exchange of a band-parallel run with ONE unit per rank: all-gather of every rank's frame block, left on
    `device` (RCCL: the GPU; gloo: the CPU).  Returns (tensor (world, block_len) float32 in rank order, total key points).
    No host staging of the gathered data - a consumer that needs another rank's rows reads them where they are
    (`blocks_to_frames(t.cpu().numpy(), ...)` builds the DataFrames on demand)."""
    import torch
    import torch.distributed as dist

    L = block_len(cap, with_zncc)
    ws = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if ws > 1 and dist.get_backend() == "nccl" else torch.device("cpu")
    mine = torch.zeros(L, dtype=torch.float32) if block is None else torch.from_numpy(np.ascontiguousarray(block[:L]))
    t_send = mine.to(device, non_blocking=False)
    if ws == 1:
        t_recv = t_send.reshape(1, L)
    else:
        t_recv = torch.empty((ws, L), dtype=torch.float32, device=device)
        dist.all_gather_into_tensor(t_recv.reshape(-1), t_send)
    total = int(t_recv[:, 0].contiguous().view(torch.int32).sum().item())      # header word 0 = n_rows (int32 bit pattern)
    return t_recv, total


def blocks_to_frames(blocks: np.ndarray, cap: int, with_zncc: bool = False):
    from .resident import ResidentPair
    return [ResidentPair._frame_from_block(b, cap, with_zncc) for b in blocks]


def match_distributed(pairs: dict, n_bands: int, x_size: int, y_size: int, conf, score: bool = False,
                      confidence_threshold: float = 0.4, device=None):
    """Match `n_bands` image pairs tile-parallel.  `pairs` maps band -> ResidentPair for the bands whose
    units this rank owns (see `units_of_rank`).  Returns the frames in reference order on every rank."""
    import torch.distributed as dist

    ws = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank() if ws > 1 else 0
    units = enumerate_units(n_bands, x_size, y_size, conf)
    local = {}
    for u in units_of_rank(units, rank, ws):
        frame = pairs[u.band].match_tile(conf, (u.x_off, u.y_off, u.x_size, u.y_size),
                                         zncc_threshold=confidence_threshold if score else None)
        if frame is not None and score:
            frame = pairs[u.band].score_frame(frame, confidence_threshold)
        local[u.index] = frame
    return gather_frames(local, len(units), int(conf.maxCorners), device=device)
