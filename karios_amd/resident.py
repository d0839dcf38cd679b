"""Device-resident matching: both full-resolution images stay in HBM, every tile of
`KLT.match` and the ZNCC scoring run on them without further host<->device image traffic.

This is the same computation as `karios_amd.matcher.KLT` + `ZNCCService` (reference
`karios/matcher/klt.py:198-349`, `karios/api/core.py:871-907`); only the place where the
pixels live differs.  Device buffers may come from `libkarios_hip` (`upload`) or be any
device pointer, e.g. a torch CUDA tensor's `data_ptr()` (`from_device_pointers`).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
from pandas import DataFrame

from . import _lib
from ._lib import Context, KariosHipError, as_image, default_context, dtype_code
from . import frames, tiling
from .ops import make_params


class DeviceBuffer:
    """Device buffer drawn from the context's pool (`Context.dev_alloc`)."""

    def __init__(self, ctx: Context, nbytes: int):
        self.ctx, self.nbytes = ctx, int(nbytes)
        self.ptr, self._cap = ctx.dev_alloc(self.nbytes)
        self.upload_in_flight = False       # an asynchronous upload into this buffer has not been joined / waited for yet

    def free(self):
        if self.ptr:
            ptr_, self.ptr = self.ptr, None
            ctx, cap, flying = self.ctx, self._cap, self.upload_in_flight
            # (a finalizer may run on another thread than the context's: the release is then left for the owning thread)
            ctx.run_or_defer(lambda: ctx.dev_release(ptr_, cap, flying))

    def __del__(self):
        try:
            self.free()
        except Exception:  # pragma: no cover
            pass

    def upload(self, arr: np.ndarray):
        """Blocking upload of a whole array."""
        a = np.ascontiguousarray(arr)
        assert a.nbytes <= self.nbytes
        self.ctx.check(self.ctx.lib.km_h2d(self.ctx.handle, C.c_void_p(self.ptr), a.ctypes.data_as(C.c_void_p), a.nbytes), "km_h2d")

    def upload_image_async(self, img: np.ndarray) -> np.ndarray:
        """Queue the upload of a 2-D image (rows may be strided views of a larger array) on the context's copy stream and
        return the array that must stay alive until the context's next call: from page-locked memory (`pinned_empty`) this
        returns at once and the copy overlaps whatever the device is computing; from pageable memory the library completes the copy
        before it returns (km_upload_async)."""
        a = as_image(img)
        assert a.shape[0] * a.shape[1] * a.itemsize <= self.nbytes
        w = a.shape[1] * a.itemsize
        self.ctx.check(self.ctx.lib.km_upload_async(self.ctx.handle, C.c_void_p(self.ptr), w, a.ctypes.data_as(C.c_void_p),
                                                    a.strides[0] if a.shape[0] > 1 else w, w, a.shape[0]), "km_upload_async")
        self.upload_in_flight = True
        return a

    def download(self, shape, dtype) -> np.ndarray:
        out = np.empty(shape, dtype)
        assert out.nbytes <= self.nbytes
        self.ctx.check(self.ctx.lib.km_d2h(self.ctx.handle, out.ctypes.data_as(C.c_void_p), C.c_void_p(self.ptr), out.nbytes), "km_d2h")
        return out


class RawFrame:
    """One tile's frame block as the device pipeline delivers it (layout of km_klt_tile_frame[_zncc]_dev): 4 int32
    {n_rows, n_init, flags, candidates} + 6*cap float32 (x0 | y0 | dx | dy | score | index bits) + `with_zncc` x cap float64
    (0 / False: none; 1 / True: zncc; 3: zncc | mutual_info_score | mi_score, `frames.block_words`)."""
    __slots__ = ("block", "cap", "with_zncc")

    def __init__(self, block: np.ndarray, cap: int, with_zncc: bool):
        self.block, self.cap, self.with_zncc = block, cap, with_zncc

    @property
    def n_rows(self) -> int:
        return int(self.block[:1].view(np.int32)[0])

    @property
    def flags(self) -> int:
        """Non-zero: the tile did not fit the fixed capacities of the synchronisation-free corner path (KM_FLAG_* of
        csrc/common.hpp) and its rows are NOT the result - repeat it through the exact path (`PendingFrame.redo`)."""
        return int(self.block[2:3].view(np.int32)[0])

    @property
    def n_candidates(self) -> int:
        return int(self.block[3:4].view(np.int32)[0])

    def to_frame(self, radial: bool = False) -> DataFrame | None:
        """The tile's DataFrame; `radial`: with the `radial error` / `angle` columns of `score_frame` already in place."""
        return ResidentPair._frame_from_block(self.block, self.cap, self.with_zncc, radial, own=True)   # (the block is this frame's private copy)


class PendingFrame:
    """A tile submitted with `ResidentPair.submit_tile`: the device is still working on its tail (LK, FB test, ZNCC, copy)
    while the caller already submits the next tile.  `wait()` (any thread) -> `RawFrame`."""

    def __init__(self, ctx: Context, ticket: int, cap: int, with_zncc: bool, redo=None):
        self.ctx, self.ticket, self.cap, self.with_zncc = ctx, ticket, cap, with_zncc
        self._raw = None
        self._redo = redo

    def redo(self) -> "RawFrame":
        """The same tile through the exact (synchronising) corner path, for a frame whose `flags` are set.  Runs device work on
        the context: call it on the thread that submits, not on a worker that only waits."""
        if self._redo is None:
            raise KariosHipError("this frame cannot be repeated")
        self._raw = self._redo()
        return self._raw

    def result(self) -> "RawFrame":
        """`wait()`, and `redo()` when the speculative path flagged the tile (submitting thread only)."""
        raw = self.wait()
        return self.redo() if raw.flags else raw

    def wait(self) -> "RawFrame":
        if self._raw is None:
            c = self.ctx
            blk, nbytes = C.c_void_p(), C.c_size_t()
            rc = c.lib.km_frame_wait(c.handle, self.ticket, C.byref(blk), C.byref(nbytes))
            if rc != 0:
                raise KariosHipError(f"km_frame_wait(ticket {self.ticket}) failed with status {rc}")
            n32 = frames.block_words(self.cap, self.with_zncc)
            pinned = np.ctypeslib.as_array(C.cast(blk, C.POINTER(C.c_float)), shape=(n32,))
            self._raw = RawFrame(pinned.copy(), self.cap, self.with_zncc)   # the pinned slot is reused 3 submissions later
        return self._raw

    def stage_ms(self) -> dict:
        """Stage spans of this frame (after `wait`, with `Context.set_profiling(True)`)."""
        c = self.ctx
        buf, n = (C.c_float * 16)(), C.c_int()
        rc = c.lib.km_frame_stage_ms(c.handle, self.ticket, buf, 16, C.byref(n))
        if rc != 0:
            raise KariosHipError(f"km_frame_stage_ms failed with status {rc}")
        return {c.lib.km_stage_name(i).decode(): float(buf[i]) for i in range(n.value)}


class PendingBatch:
    """Units submitted together with `submit_units` (km_klt_units_frame_submit): ONE device pipeline for all of them.
    `wait()` (any thread) -> the units' `RawFrame`s in submission order; `redo(i)` repeats unit i alone through the exact path."""

    def __init__(self, ctx: Context, ticket: int, cap: int, with_zncc, redos):
        import threading
        self.ctx, self.ticket, self.cap, self.with_zncc = ctx, ticket, cap, with_zncc
        self._redos = redos
        self._raws = None
        self._submitter = threading.get_ident()

    def __len__(self):
        return len(self._redos)

    def wait(self) -> "list[RawFrame]":
        if self._raws is None:
            import threading
            c = self.ctx
            if threading.get_ident() == self._submitter:
                c.flush(self.ticket)        # ("units_pipeline": the submitting thread waits and no further submission came - the deferred tail goes now;
                                 #  a worker thread just waits: the submitting thread submits the next batch or flushes - FrameStream does)
            blk, nbytes = C.c_void_p(), C.c_size_t()
            rc = c.lib.km_frame_wait(c.handle, self.ticket, C.byref(blk), C.byref(nbytes))
            if rc != 0:
                raise KariosHipError(f"km_frame_wait(ticket {self.ticket}) failed with status {rc}")
            n32 = frames.block_words(self.cap, self.with_zncc)
            n = len(self._redos)
            pinned = np.ctypeslib.as_array(C.cast(blk, C.POINTER(C.c_float)), shape=(n * n32,))
            own = pinned.copy()                                   # the pinned slot is reused 3 submissions later
            self._raws = [RawFrame(own[k * n32:(k + 1) * n32], self.cap, self.with_zncc) for k in range(n)]
        return self._raws

    def redo(self, i: int) -> "RawFrame":
        """Unit i through the exact (synchronising) corner path (submitting thread only)."""
        self._raws[i] = self._redos[i]()
        return self._raws[i]

    def stage_ms(self) -> dict:
        """Stage spans of the whole batch (after `wait`, with `Context.set_profiling(True)`)."""
        return PendingFrame.stage_ms(self)


def submit_units(units, conf, zncc_threshold=None, mutual_info: bool = False) -> "PendingBatch | None":
    """`ResidentPair.submit_tile` for up to 16 independent units at once - `units` = [(pair, box | None, origin | None), ...], all on
    ONE context, one pixel type, all with a user mask or none: the tiles of `KLT.match` (klt.py:220-253), of one pair or of several bands.  Every
    dense kernel, the corner-selection chain, LK, the frame stage and the scores are launched ONCE for all units
    (csrc/api_units.hip); the frames are the unit-by-unit frames bit for bit.  Returns None when the batch form does not cover the
    case (the library answers KM_E_UNSUPPORTED: maxCorners 0, a unit narrower than 512 columns ...) or the
    units do not share a context / pixel type, or only some carry a user mask - submit them one by one then."""
    if not units or len(units) > _lib.UNITS_PER_SUBMISSION:
        return None
    if conf.laplacian_kernel_size == "auto" or conf.laplacian_invert_polarity == "auto" or getattr(conf, "outliers_filtering", False) or conf.maxCorners <= 0:
        return None
    first = units[0][0]
    c = first.ctx
    if any(p.ctx is not c or p.code != first.code or bool(p.mask_ptr) != bool(first.mask_ptr) or p.no_data_mon != first.no_data_mon
           or p.no_data_ref != first.no_data_ref for p, _, _ in units):
        return None
    with_zncc = zncc_threshold is not None
    n_scores = 0 if not with_zncc else (3 if mutual_info else 1)
    prm = ResidentPair._params(conf)
    cap = prm.max_corners
    arr = (_lib.KmUnit * len(units))()
    redos = []
    for k, (pair, box, origin) in enumerate(units):
        bx_off, by_off, bx, by, off = pair._box(box)
        x_off, y_off = origin if origin is not None else (bx_off, by_off)
        es = pair.dtype.itemsize
        u = arr[k]
        u.d_ref, u.d_mon, u.sref, u.smon, u.H, u.W = pair.ref_ptr + off * es, pair.mon_ptr + off * es, pair.x_size, pair.x_size, by, bx
        u.x_off, u.y_off = float(x_off), float(y_off)
        if with_zncc:
            u.d_ref_full, u.d_mon_full, u.sref_f, u.smon_f, u.Hf, u.Wf = pair.ref_ptr, pair.mon_ptr, pair.x_size, pair.x_size, pair.y_size, pair.x_size
        if pair.window is not None:
            u.win_ox, u.win_oy, u.win_H, u.win_W = (int(v) for v in pair.window)
        if pair.mask_ptr:                     # the user mask of the box (klt.py:258-266): a uint8 raster of the pair's shape
            u.d_mask, u.smask = pair.mask_ptr + off, pair.x_size

        def exact(pair=pair, box=box, origin=(x_off, y_off)):
            before = c.get_option("speculative", int(os.environ.get("KARIOS_HIP_SPECULATIVE", "1") or 0))
            c.set_option("speculative", 0)
            sink = c.frame_sink
            if sink[0]:
                c.set_frame_sink(None)
            try:
                raw = pair.match_tile_raw(conf, box, zncc_threshold, origin=origin, mutual_info=mutual_info)
                return RawFrame(raw.block.copy(), raw.cap, raw.with_zncc)      # (match_tile_raw's blocks rotate through a ring of three)
            finally:
                c.set_option("speculative", before)
                if sink[0]:
                    c.set_frame_sink(*sink)
        redos.append(exact)
    nr = C.byref(C.c_double(float(first.no_data_ref))) if first.no_data_ref is not None else None
    nm = C.byref(C.c_double(float(first.no_data_mon))) if first.no_data_mon is not None else None
    ticket = C.c_int(-1)
    with first._frame_mi(n_scores == 3):
        rc = c.lib.km_klt_units_frame_submit(c.handle, arr, len(units), first.code, nr, nm, C.byref(prm), float(zncc_threshold or 0.0), cap, C.byref(ticket))
    if rc == _lib.E_UNSUPPORTED:
        return None
    c.check(rc, "km_klt_units_frame_submit")
    return PendingBatch(c, ticket.value, cap, n_scores, redos)


class ResidentPair:
    """A monitored / reference image pair (plus optional user mask) resident in HBM."""

    def __init__(self, ctx: Context | None, mon_ptr: int, ref_ptr: int, dtype, y_size: int, x_size: int,
                 mask_ptr: int | None = None, no_data_mon=None, no_data_ref=None, owned=()):
        self.ctx = ctx if ctx is not None else default_context()
        self.mon_ptr, self.ref_ptr, self.mask_ptr = int(mon_ptr), int(ref_ptr), (int(mask_ptr) if mask_ptr else None)
        self.dtype = np.dtype(dtype)
        self.code = dtype_code(np.empty(0, self.dtype))
        self.y_size, self.x_size = int(y_size), int(x_size)
        self.no_data_mon, self.no_data_ref = no_data_mon, no_data_ref
        self._owned = list(owned)
        # (ox, oy, image y_size, image x_size) when the buffers are only a window of a larger image (km_set_image_window):
        # key points are then image coordinates everywhere (tile `origin`s, zncc / mutual_info arguments)
        self.window = None
        self._out = None
        self._host_frame = None

    def __del__(self):
        # a pair that was uploaded (e.g. prefetched) but never used still holds an upload ticket: give its slot and event back
        # (never from the finalizer's thread unless it is the context's own: km_upload_join mutates the context)
        try:
            t = self.__dict__.pop("_upload_ticket", -1)
            ctx = self.ctx
            if t >= 0 and getattr(ctx, "handle", None):
                ctx.run_or_defer(lambda: ctx.handle and ctx.lib.km_upload_join(ctx.handle, t))
        except Exception:  # pragma: no cover - interpreter shutdown
            pass

    @classmethod
    def upload(cls, mon: np.ndarray, ref: np.ndarray, mask: np.ndarray | None = None, ctx: Context | None = None,
               no_data_mon=None, no_data_ref=None) -> "ResidentPair":
        ctx = ctx if ctx is not None else default_context()
        mon, ref = np.asarray(mon), np.asarray(ref)
        if mon.shape != ref.shape or mon.dtype != ref.dtype or mon.ndim != 2:
            raise KariosHipError("ResidentPair: mon/ref must be 2-D arrays of equal shape and dtype")
        dtype_code(mon)                                   # unsupported pixel types fail before anything is allocated
        bm, br = DeviceBuffer(ctx, mon.nbytes), DeviceBuffer(ctx, ref.nbytes)
        # the copies travel on the context's copy stream; the pair's first device call waits for them ON THE DEVICE, so a
        # pair uploaded from page-locked memory while the previous one is being matched costs no time
        keep = [br.upload_image_async(ref), bm.upload_image_async(mon)]
        owned, mptr = [bm, br], None
        if mask is not None:
            mk = np.asarray(mask)
            if mk.dtype != np.uint8:
                mk = mk.astype(np.uint8)
            if mk.shape != mon.shape:
                raise KariosHipError("ResidentPair: mask shape differs from the images'")
            bk = DeviceBuffer(ctx, mk.nbytes)
            keep.append(bk.upload_image_async(mk))
            owned.append(bk)
            mptr = bk.ptr
        pair = cls(ctx, bm.ptr, br.ptr, mon.dtype, mon.shape[0], mon.shape[1], mptr, no_data_mon, no_data_ref, owned)
        pair._upload_sources = keep
        ticket = C.c_int(-1)
        ctx.check(ctx.lib.km_upload_mark(ctx.handle, C.byref(ticket)), "km_upload_mark")
        pair._upload_ticket = ticket.value      # joined by the pair's first device call: work queued for OTHER pairs meanwhile does not wait
        return pair

    @classmethod
    def from_windows(cls, mon, ref, mask=None, ctx: Context | None = None, no_data_mon=None, no_data_ref=None) -> "ResidentPair":
        """Pair from boxes that already live in HBM (`karios_amd.core.DeviceWindow`, what `DeviceRasterImage.read` returns):
        dense device copies (device-to-device, nothing crosses PCIe) owned by the pair."""
        import torch
        m, r = mon.view.contiguous(), ref.view.contiguous()
        if m.data_ptr() == mon.view.data_ptr():
            m = m.clone()                                   # the pair owns its pixels like an uploaded one
        if r.data_ptr() == ref.view.data_ptr():
            r = r.clone()
        k = None
        if mask is not None:
            from .core.image import DeviceWindow
            k = mask.view.contiguous() if isinstance(mask, DeviceWindow) else torch.as_tensor(np.ascontiguousarray(mask, np.uint8), device=m.device)
        torch.cuda.synchronize(m.device)                    # the library runs on its own stream
        return cls(ctx, m.data_ptr(), r.data_ptr(), mon.dtype, m.shape[0], m.shape[1], None if k is None else k.data_ptr(),
                   no_data_mon, no_data_ref, owned=[m, r, k])

    @classmethod
    def from_device_pointers(cls, mon_ptr: int, ref_ptr: int, dtype, y_size: int, x_size: int, ctx: Context | None = None,
                             mask_ptr: int | None = None, no_data_mon=None, no_data_ref=None, keepalive=()) -> "ResidentPair":
        """Wrap existing device memory (row-major, dense rows), e.g. torch tensors' data_ptr().
        The library works on its own HIP stream: the producer of these buffers must have completed
        (e.g. `torch.cuda.synchronize()`) before the first call."""
        return cls(ctx, mon_ptr, ref_ptr, dtype, y_size, x_size, mask_ptr, no_data_mon, no_data_ref, keepalive)

    # ------------------------------------------------------------------ KLT
    def _outputs(self, cap: int):
        need = 3 * cap * 2 * 4 + 16
        if self._out is None or self._out.nbytes < need:
            self._out = DeviceBuffer(self.ctx, need)
        base = self._out.ptr
        return base, base + cap * 8, base + 2 * cap * 8, base + 3 * cap * 8

    def _windowed(self):
        """Context manager: the ZNCC / MI kernels of the calls inside see this pair's image window."""
        from contextlib import contextmanager

        @contextmanager
        def scope():
            if self.window is None:
                yield
                return
            c = self.ctx
            c.check(c.lib.km_set_image_window(c.handle, *(int(v) for v in self.window)), "km_set_image_window")
            try:
                yield
            finally:
                c.lib.km_set_image_window(c.handle, 0, 0, 0, 0)
        return scope()

    def _ready(self):
        """Before the first kernel on this pair: its uploads (and only its uploads) must have landed."""
        self.ctx.drain()
        t = self.__dict__.pop("_upload_ticket", -1)
        if t >= 0:
            self.ctx.check(self.ctx.lib.km_upload_join(self.ctx.handle, t), "km_upload_join")
            for buf in self._owned:          # from here on the compute stream is ordered behind the copies: a context sync covers them
                if isinstance(buf, DeviceBuffer):
                    buf.upload_in_flight = False

    def _box(self, box):
        """-> (x_off, y_off, x_size, y_size, element offset of the box origin) of a validated box (None = whole pair)."""
        self._ready()
        x_off, y_off, bx, by = box if box is not None else (0, 0, self.x_size, self.y_size)
        if x_off < 0 or y_off < 0 or x_off + bx > self.x_size or y_off + by > self.y_size or bx <= 0 or by <= 0:
            raise KariosHipError(f"box {box} outside the {self.x_size}x{self.y_size} image")
        return x_off, y_off, bx, by, y_off * self.x_size + x_off

    def _image_args(self, off):
        """ctypes arguments shared by the tile entry points: ref, mon pointers of the box, the mask / no-data pointers."""
        es = self.dtype.itemsize
        mask = C.c_void_p(self.mask_ptr + off) if self.mask_ptr else None
        nr = C.byref(C.c_double(float(self.no_data_ref))) if self.no_data_ref is not None else None
        nm = C.byref(C.c_double(float(self.no_data_mon))) if self.no_data_mon is not None else None
        return C.c_void_p(self.ref_ptr + off * es), C.c_void_p(self.mon_ptr + off * es), mask, nr, nm

    @staticmethod
    def _params(conf, ksizes=None, invert_mon=None):
        mon_k, ref_k = ksizes if ksizes is not None else tiling.kernel_sizes(conf.laplacian_kernel_size)
        invert = bool(conf.laplacian_invert_polarity) if invert_mon is None else bool(invert_mon)
        return make_params(conf, mon_k, ref_k, invert)

    def track_tile(self, conf, box=None, mon_ksize=None, ref_ksize=None, invert_mon=False, out_ptrs=None):
        """km_klt_tile_dev on one box (x_off, y_off, x_size, y_size) of the resident pair.
        -> (status, (p0, p1, p0r) | None) like `ops.klt_tile`; with `out_ptrs` (p0, p1, p0r, n device
        pointers, capacity) the results stay on the device and only (status, n) is returned."""
        c = self.ctx
        _, _, bx, by, off = self._box(box)
        prm = self._params(conf, None if mon_ksize is None or ref_ksize is None else (mon_ksize, ref_ksize), invert_mon)
        cap = prm.max_corners if prm.max_corners > 0 else max(1, (bx * by) // 4)
        if out_ptrs is not None:
            d0, d1, d2, dn, ocap = out_ptrs
            if ocap < cap:
                raise KariosHipError("track_tile: output capacity below maxCorners")
        else:
            d0, d1, d2, dn = self._outputs(cap)
        ref, mon, mask, nr, nm = self._image_args(off)
        c.check(c.lib.km_klt_tile_dev(c.handle, ref, mon, self.code, by, bx, self.x_size, self.x_size, mask, self.x_size, nr, nm,
                                      C.byref(prm), C.c_void_p(d0), C.c_void_p(d1), C.c_void_p(d2), cap, C.c_void_p(dn)), "km_klt_tile_dev")
        st = c.stats()  # valid_pixels is known on the host as soon as the call returns
        if st.valid_pixels == 0:
            return "no_valid_pixels", None
        nbuf = np.empty(1, np.int32)
        c.check(c.lib.km_d2h(c.handle, nbuf.ctypes.data_as(C.c_void_p), C.c_void_p(dn), 4), "km_d2h")
        n = int(nbuf[0])
        if n == 0:
            return "no_features", None
        if out_ptrs is not None:
            return "ok", n
        pts = np.empty((3, n, 2), np.float32)
        for i, d in enumerate((d0, d1, d2)):
            c.check(c.lib.km_d2h(c.handle, pts[i].ctypes.data_as(C.c_void_p), C.c_void_p(d), n * 8), "km_d2h")
        return "ok", tuple(p.reshape(-1, 1, 2) for p in pts)

    def match_tile(self, conf, box=None, zncc_threshold=None, ksizes=None, invert_mon=None, origin=None, mutual_info: bool = False) -> DataFrame | None:
        """One tile of `KLT.match` (reference klt.py:236-349) on resident data with fixed Laplacian kernel sizes and
        polarity: `ksizes` (mon, ref) / `invert_mon` default to the configuration's (its 'auto' values are the caller's
        business: `karios_amd.matcher.KLT`).  `origin` (x, y) is added to the key points (default: the box offset).
        With `zncc_threshold` the frame also carries the `zncc_score` column of `_handle_klt_results` (core.py:876-893)
        computed in the same device call."""
        if ksizes is None and conf.laplacian_kernel_size == "auto" or invert_mon is None and conf.laplacian_invert_polarity == "auto":
            raise KariosHipError("ResidentPair.match_tile: 'auto' modes are resolved by karios_amd.matcher.KLT")
        x_off, y_off = origin if origin is not None else ((box[0], box[1]) if box is not None else (0, 0))
        if getattr(conf, "outliers_filtering", False):
            # the iterative sigma clip (klt.py:52-71) needs numpy's float32 statistics: FB test and ordering on the host
            mon_k, ref_k = ksizes if ksizes is not None else tiling.kernel_sizes(conf.laplacian_kernel_size)
            invert = bool(conf.laplacian_invert_polarity) if invert_mon is None else bool(invert_mon)
            status, tracks = self.track_tile(conf, box, mon_k, ref_k, invert)
            if status != "ok":
                return None
            cols, n_init = frames.track_columns(*tracks)
            points = frames.assemble(cols, x_off, y_off, clip_outliers=True)
            points.attrs["Ninit"] = n_init
            if zncc_threshold is not None:
                keep = points["score"].to_numpy() >= zncc_threshold
                z = np.full(len(points), np.nan)
                if keep.any():
                    z[keep] = self.zncc(*(points[c].to_numpy()[keep] for c in ("x0", "y0", "dx", "dy")))
                points["zncc_score"] = z
            return points
        return self._match_tile_device_frame(conf, box, x_off, y_off, zncc_threshold, ksizes=ksizes, invert_mon=invert_mon, mutual_info=mutual_info)

    def _frame_mi(self, on: bool):
        """Context manager: frames scored inside also carry `mutual_info_score` / `mi_score` (km_set_option "frame_mi")."""
        from contextlib import contextmanager

        @contextmanager
        def scope():
            if not on:
                yield
                return
            self.ctx.set_option("frame_mi", 1)
            try:
                yield
            finally:
                self.ctx.set_option("frame_mi", 0)
        return scope()

    def _match_tile_device_frame(self, conf, box, x_off, y_off, zncc_threshold=None, build_frame=True, ksizes=None,
                                 invert_mon=None, mutual_info: bool = False) -> DataFrame | None:
        """Tile pipeline + FB test + score + (x0, y0) ordering (+ ZNCC [+ MI / NMI] of the confident rows) on the device, one D2H
        copy of the finished frame."""
        c = self.ctx
        n_scores = 0 if zncc_threshold is None else (3 if mutual_info else 1)
        _, _, bx, by, off = self._box(box)
        prm = self._params(conf, ksizes, invert_mon)
        cap = prm.max_corners if prm.max_corners > 0 else max(1, (bx * by) // 4)
        if self._host_frame is None or self._host_frame.size < frames.block_words(cap, 3):
            self._host_frame = np.empty(frames.block_words(cap, 3), np.float32)
        buf = self._host_frame
        ref, mon, mask, nr, nm = self._image_args(off)
        if zncc_threshold is None:
            c.check(c.lib.km_klt_tile_frame_dev(c.handle, ref, mon, self.code, by, bx, self.x_size, self.x_size, mask, self.x_size, nr, nm,
                                                C.byref(prm), float(x_off), float(y_off), buf.ctypes.data_as(C.c_void_p), cap),
                    "km_klt_tile_frame_dev")
        else:
            with self._windowed(), self._frame_mi(n_scores == 3):
                c.check(c.lib.km_klt_tile_frame_zncc_dev(c.handle, ref, mon, self.code, by, bx, self.x_size, self.x_size, mask, self.x_size, nr, nm,
                                                         C.byref(prm), float(x_off), float(y_off), C.c_void_p(self.ref_ptr),
                                                         C.c_void_p(self.mon_ptr), self.y_size, self.x_size, self.x_size, self.x_size,
                                                         float(zncc_threshold), buf.ctypes.data_as(C.c_void_p), cap), "km_klt_tile_frame_zncc_dev")
        if not build_frame:
            return None
        frame = frames.block_to_frame(buf, cap, n_scores)
        if frame is not None:
            frame.attrs["Ninit"] = int(buf[:2].view(np.int32)[1])
        return frame

    def match_tile_auto_ksize(self, conf, box=None, invert_mon: bool = False, candidates=tiling.AUTO_KSIZE_CANDIDATES, origin=None):
        """The Laplacian kernel-size search of the reference (klt.py:465-545) for one tile of the resident pair in ONE
        device call: every Laplacian, pyramid and corner list is built once on the device and shared by the
        len(candidates)^2 tracker runs; the winner is the (mon, ref) pair with the highest inlier ratio, first in
        (mon outer, ref inner) order on ties.
        -> (frame | None, {(mon_k, ref_k): inlier ratio}, (mon_k, ref_k) | None, Ninit).  No outlier filtering."""
        if getattr(conf, "outliers_filtering", False):
            raise KariosHipError("match_tile_auto_ksize: the sigma clip changes the inlier counts - use karios_amd.matcher.KLT")
        c = self.ctx
        bx_off, by_off, bx, by, off = self._box(box)
        prm = make_params(conf, 1, 1, bool(invert_mon))
        cap = prm.max_corners if prm.max_corners > 0 else max(1, (bx * by) // 4)
        buf = np.empty(4 + 6 * cap, np.float32)
        ks = np.ascontiguousarray(candidates, np.int32)
        nk = len(ks)
        ratios = np.zeros(nk * nk, np.float64)
        best = np.zeros(2, np.int32)
        ref, mon, mask, nr, nm = self._image_args(off)
        x_off, y_off = origin if origin is not None else (bx_off, by_off)
        c.check(c.lib.km_klt_auto_ksize_frame_dev(c.handle, ref, mon, self.code, by, bx, self.x_size, self.x_size, mask, self.x_size, nr, nm,
                                                  C.byref(prm), ks.ctypes.data_as(C.c_void_p), nk, float(x_off), float(y_off),
                                                  buf.ctypes.data_as(C.c_void_p), cap, ratios.ctypes.data_as(C.c_void_p),
                                                  best.ctypes.data_as(C.c_void_p)), "km_klt_auto_ksize_frame_dev")
        scores = {(int(ks[i]), int(ks[j])): float(ratios[i * nk + j]) for i in range(nk) for j in range(nk)}
        if best[0] < 0:
            return None, scores, None, 0
        n_init = int(buf[:2].view(np.int32)[1])
        return frames.block_to_frame(buf, cap), scores, (int(best[0]), int(best[1])), n_init

    def match_tile_raw(self, conf, box=None, zncc_threshold=None, origin=None, mutual_info: bool = False) -> "RawFrame":
        """GPU half of `match_tile`: runs the device pipeline and returns the raw frame block (a private copy), leaving
        the pandas half to `RawFrame.to_frame()` - which may run in another thread while this thread already drives the
        next tile (ctypes releases the GIL inside the library).  Fixed kernel size / polarity, no outlier filtering."""
        if conf.laplacian_kernel_size == "auto" or conf.laplacian_invert_polarity == "auto" or getattr(conf, "outliers_filtering", False):
            raise KariosHipError("ResidentPair.match_tile_raw: 'auto' modes and outlier filtering need ResidentPair.match_tile / matcher.KLT")
        x_off, y_off = origin if origin is not None else ((box[0], box[1]) if box is not None else (0, 0))
        with_zncc = 0 if zncc_threshold is None else (3 if mutual_info else 1)
        # host blocks rotate through a ring of three: the previous block may still be read by the host half of the pipeline
        ring = self.__dict__.setdefault("_raw_ring", [None, None, None])
        slot = self.__dict__.get("_raw_slot", 0)
        self._raw_slot = (slot + 1) % len(ring)
        self._host_frame = ring[slot]
        self._match_tile_device_frame(conf, box, x_off, y_off, zncc_threshold, build_frame=False, mutual_info=mutual_info)
        bx, by = (box[2], box[3]) if box is not None else (self.x_size, self.y_size)
        cap = conf.maxCorners if conf.maxCorners > 0 else max(1, (bx * by) // 4)
        block, ring[slot], self._host_frame = self._host_frame, self._host_frame, None
        return RawFrame(block[:frames.block_words(cap, with_zncc)], cap, with_zncc)

    def submit_tile(self, conf, box=None, zncc_threshold=None, origin=None, mutual_info: bool = False) -> PendingFrame:
        """Asynchronous `match_tile_raw` (km_klt_tile_frame_submit): returns when the tile's last kernel and the copy of
        its frame block are enqueued, so the next `submit_tile` queues its dense stages right behind them - no GPU idle
        time between tiles.  Up to three tiles may be pending; `PendingFrame.wait()` may run in another thread."""
        if conf.laplacian_kernel_size == "auto" or conf.laplacian_invert_polarity == "auto" or getattr(conf, "outliers_filtering", False):
            raise KariosHipError("ResidentPair.submit_tile: 'auto' modes and outlier filtering need ResidentPair.match_tile / matcher.KLT")
        c = self.ctx
        bx_off, by_off, bx, by, off = self._box(box)
        prm = self._params(conf)
        cap = prm.max_corners if prm.max_corners > 0 else max(1, (bx * by) // 4)
        ref, mon, mask, nr, nm = self._image_args(off)
        with_zncc = zncc_threshold is not None
        n_scores = 0 if not with_zncc else (3 if mutual_info else 1)
        ticket = C.c_int(-1)
        x_off, y_off = origin if origin is not None else (bx_off, by_off)
        with self._windowed(), self._frame_mi(n_scores == 3):
            c.check(c.lib.km_klt_tile_frame_submit(c.handle, ref, mon, self.code, by, bx, self.x_size, self.x_size, mask, self.x_size, nr, nm,
                                                   C.byref(prm), float(x_off), float(y_off),
                                                   C.c_void_p(self.ref_ptr) if with_zncc else None, C.c_void_p(self.mon_ptr) if with_zncc else None,
                                                   self.y_size, self.x_size, self.x_size, self.x_size, float(zncc_threshold or 0.0), cap,
                                                   C.byref(ticket)), "km_klt_tile_frame_submit")
        def exact():
            before = c.get_option("speculative", int(os.environ.get("KARIOS_HIP_SPECULATIVE", "1") or 0))
            c.set_option("speculative", 0)
            # the frame sink (km_set_frame_sink) now points at the slot of a NEWER unit: the repeat must not overwrite that block
            # (the all-gather then shipped the redone block of step s - D in the place of step s's own); the caller
            # takes the repeated block from the returned RawFrame
            sink = c.frame_sink
            if sink[0]:
                c.set_frame_sink(None)
            try:
                raw = self.match_tile_raw(conf, box, zncc_threshold, origin=(x_off, y_off), mutual_info=mutual_info)
                return RawFrame(raw.block.copy(), raw.cap, raw.with_zncc)      # (match_tile_raw's blocks rotate through a ring of three)
            finally:
                c.set_option("speculative", before)
                if sink[0]:
                    c.set_frame_sink(*sink)
        return PendingFrame(c, ticket.value, cap, n_scores, redo=exact)

    def match_pipelined(self, conf, boxes=None, zncc_threshold=None, host_stage=None, with_empty: bool = False, split_small_grids: bool = True):
        """`match` as a pipeline (`karios_amd.stream.FrameStream`): tile i+1 is submitted to the device (`submit_tile`) while a
        worker thread waits for tile i and builds its DataFrame.  `host_stage(frame)` (e.g. `score_frame`) runs on the CALLING
        thread when the frame is collected: a context is not thread-safe, and a stage that calls back into the library (ZNCC /
        MI of rows the device call did not score) must not run beside `submit_tile`.  Yields the frames in tile order, like
        `KLT.match`."""
        from .stream import FrameStream
        if boxes is None:
            boxes = tiling.tile_grid(self.x_size, self.y_size, conf.tile_size, conf.xStart)
        stage = None if host_stage is None else (lambda frame, _pair: host_stage(frame))
        with FrameStream(zncc_threshold, depth=1, host_stage=stage, score_columns=False) as stream:
            # the tiles travel as batched submissions (one set of device launches per <= 16 tiles) where the batch form covers them.  A grid
            # of 8 .. 16 tiles goes as TWO submissions: the device pipelines them (csrc/api_units.hip) and the first half's DataFrames are
            # built while the second half is on the device - as one submission every frame waited for the last tile (tools/investigations/e2e_shape_probe.py:
            # 16 tiles of 3000^2 15.0 -> 14.45 ms per pair; four tiles gain nothing: 2.45 against 2.52)
            units = [(self, box, None) for box in boxes]
            halves = [units[:len(units) // 2], units[len(units) // 2:]] if split_small_grids and 8 <= len(units) <= 16 else [units]
            for part in halves:
                for done in stream.submit_many(part, conf):
                    if done.frame is not None or with_empty:          # (with_empty: None for a tile without valid pixels / corners)
                        yield done.frame
            for done in stream.drain():
                if done.frame is not None or with_empty:
                    yield done.frame

    _frame_from_block = staticmethod(frames.block_to_frame)

    def last_block(self, cap: int, with_zncc: bool) -> np.ndarray:
        """The raw frame block of the last `match_tile` call: 4 int32 header + 6*cap float32 (+ cap float64), the unit of
        the multi-GPU gather (no pandas round trip)."""
        return self._host_frame[:frames.block_words(cap, with_zncc)]

    def match(self, conf):
        """All tiles in the reference order (x outer, y inner; klt.py:220-232)."""
        for box in tiling.tile_grid(self.x_size, self.y_size, conf.tile_size, conf.xStart):
            frame = self.match_tile(conf, box)
            if frame is not None:
                yield frame

    # ------------------------------------------------------------------ ZNCC
    def _scratch(self, n_keypoints: int, n_outputs: int):
        """Page-locked staging for key-point columns in and scores out: the kernels read / write it DIRECTLY over PCIe (a few
        hundred KB), so neither direction queues a copy behind an image upload that is in flight on the DMA engine.
        -> (float32 view of 4 * n key-point values, float64 view of n_outputs * n scores, their addresses)."""
        from ._lib import pinned_empty
        need = n_keypoints * (4 * 4 + n_outputs * 8)
        # one grow-only buffer per CONTEXT: releasing page-locked memory synchronises with every copy in flight, so a pair
        # that owned its staging would stall the pipeline whenever it is dropped while the next pair's upload travels
        raw = self.ctx.__dict__.get("_kp_staging")
        if raw is None or raw.nbytes < need:
            raw = self.ctx.__dict__["_kp_staging"] = pinned_empty(need + need // 2 + 64, np.uint8, self.ctx)
        scores = raw[:n_outputs * n_keypoints * 8].view(np.float64)
        kp = raw[n_outputs * n_keypoints * 8:n_outputs * n_keypoints * 8 + 16 * n_keypoints].view(np.float32)
        return kp, scores, kp.ctypes.data, scores.ctypes.data

    def zncc(self, x0, y0, dx, dy) -> np.ndarray:
        """ZNCCService.compute_zncc values (zncc_service.py:186-238) for key points given in image coordinates."""
        self._ready()
        c = self.ctx
        cols = [np.ascontiguousarray(v, np.float32) for v in (x0, y0, dx, dy)]
        n = len(cols[0])
        if n == 0:
            return np.empty(0, np.float64)
        kp, scores, f, o = self._scratch(n, 1)
        np.concatenate(cols, out=kp)
        with self._windowed():
            c.check(c.lib.km_zncc_batch_dev(c.handle, C.c_void_p(self.ref_ptr), C.c_void_p(self.mon_ptr), self.code, self.y_size, self.x_size,
                                            self.y_size, self.x_size, self.x_size, self.x_size, C.c_void_p(f), C.c_void_p(f + 4 * n),
                                            C.c_void_p(f + 8 * n), C.c_void_p(f + 12 * n), n, C.c_void_p(o)), "km_zncc_batch_dev")
        c.sync()
        return scores.copy()

    def mutual_info(self, x0, y0, dx, dy):
        """(mutual_info_score, mi_score) per key point on the resident images: `MutualInfoService.compute_mutual_info`
        (mutual_info_service.py:73-130) and `ZNCCService.compute_mi` (zncc_service.py:240-287)."""
        self._ready()
        c = self.ctx
        cols = [np.ascontiguousarray(v, np.float32) for v in (x0, y0, dx, dy)]
        n = len(cols[0])
        if n == 0:
            return np.empty(0, np.float64), np.empty(0, np.float64)
        # the two scores come out of ONE kernel run; the reference asks for them in two separate service calls on the same
        # key points (core.py:894-907), so the last result is remembered
        digest = b"".join(v.tobytes() for v in cols)          # the key points themselves (16 B each), compared byte for byte
        memo = self.__dict__.get("_mi_memo")
        if memo is not None and memo[0] == digest:
            return memo[1].copy(), memo[2].copy()
        kp, scores, f, o = self._scratch(n, 2)
        np.concatenate(cols, out=kp)
        with self._windowed():
            c.check(c.lib.km_mi_batch_dev(c.handle, C.c_void_p(self.ref_ptr), C.c_void_p(self.mon_ptr), self.code, self.y_size, self.x_size,
                                          self.y_size, self.x_size, self.x_size, self.x_size, C.c_void_p(f), C.c_void_p(f + 4 * n),
                                          C.c_void_p(f + 8 * n), C.c_void_p(f + 12 * n), n, C.c_void_p(o), C.c_void_p(o + 8 * n)),
                    "km_mi_batch_dev")
        c.sync()
        st, nmi = scores[:n].copy(), scores[n:].copy()
        self._mi_memo = (digest, st, nmi)
        return st.copy(), nmi.copy()

    def score_frame(self, frame: DataFrame, confidence_threshold: float = 0.4, mutual_info: bool = False) -> DataFrame:
        """`_handle_klt_results` numeric columns (core.py:872-907): radial error, angle, the ZNCC of the rows with
        score >= confidence_threshold (NaN elsewhere) and, with `mutual_info`, the `mutual_info_score` / `mi_score`
        columns of the same rows."""
        frame = frames.radial_angle_columns(frame)
        dx, dy, score = frame["dx"].to_numpy(), frame["dy"].to_numpy(), frame["score"].to_numpy()
        keep = score >= confidence_threshold
        if "zncc_score" not in frame.columns:   # else: already scored on the device (match_tile(..., zncc_threshold=...))
            z = np.full(len(frame), np.nan, np.float64)
            if keep.any():
                z[keep] = self.zncc(frame["x0"].to_numpy()[keep], frame["y0"].to_numpy()[keep], dx[keep], dy[keep])
            frame["zncc_score"] = z
        if mutual_info and "mutual_info_score" not in frame.columns:     # else: scored by the device call that produced the frame
            st = np.full(len(frame), np.nan, np.float64)
            nmi = np.full(len(frame), np.nan, np.float64)
            if keep.any():
                st[keep], nmi[keep] = self.mutual_info(frame["x0"].to_numpy()[keep], frame["y0"].to_numpy()[keep], dx[keep], dy[keep])
            frame["mutual_info_score"] = st
            frame["mi_score"] = nmi
        return frame

    # ------------------------------------------------------------------ large offset
    def phase_offset(self) -> np.ndarray:
        """LargeOffsetMatcher.match() on resident data (large_offset.py:39): [row, col]."""
        self._ready()
        c = self.ctx
        out = (C.c_double * 2)()
        c.check(c.lib.km_phase_shift_dev(c.handle, C.c_void_p(self.mon_ptr), C.c_void_p(self.ref_ptr), self.code, self.y_size, self.x_size,
                                         self.x_size, self.x_size, out), "km_phase_shift_dev")
        return np.array([out[0], out[1]], np.float64)

    def shifted_monitored(self, y_off: int, x_off: int) -> "ResidentPair":
        """shift_image(mon, y_off, x_off) on the device (image.py:70-101); returns a new pair sharing ref."""
        self._ready()
        c = self.ctx
        buf = DeviceBuffer(c, self.y_size * self.x_size * self.dtype.itemsize)
        c.check(c.lib.km_shift_image_dev(c.handle, C.c_void_p(self.mon_ptr), self.dtype.itemsize, self.y_size, self.x_size, self.x_size,
                                         int(y_off), int(x_off), C.c_void_p(buf.ptr)), "km_shift_image_dev")
        return ResidentPair(c, buf.ptr, self.ref_ptr, self.dtype, self.y_size, self.x_size, self.mask_ptr, self.no_data_mon,
                            self.no_data_ref, owned=[buf, self])


# ---------------------------------------------------------------------------- pairs shared between the matcher services
# `KariosAPI._handle_klt_results` (core.py:888-907) hands the SAME two rasters to KLT.match, ZNCCService.compute_zncc,
# MutualInfoService.compute_mutual_info and ZNCCService.compute_mi, one after the other; each of them uploading both images
# again would cost 3 x 482 MB over PCIe for a Sentinel-2 pair.  The services therefore look the images up here first.
#
# A resident copy must never stand in for host pixels that have changed since the upload.  Reading the host arrays again to verify
# them costs more than uploading them (a 482 MB pair crosses PCIe in 8.5 ms; a host-side checksum of it takes longer), so validity is
# a TOKEN, not a fingerprint:
#   * an entry is keyed on the exact buffers that were uploaded (address, shape, strides, dtype) and keeps them alive, so their
#     addresses cannot be recycled by another array meanwhile;
#   * while an entry exists its host arrays (and the arrays they are views of) are READ-ONLY: an in-place edit raises numpy's
#     "assignment destination is read-only" instead of silently scoring against stale HBM data; a lookup through an array that is
#     writeable again (somebody lifted the guard) is a miss;
#   * the raster objects carry the token's other half: `NumpyRasterImage.clear_cache()` - and `forget_shared_pairs()` - drop the
#     entry and give write access back; a `GdalRasterImage.clear_cache()` drops its array, so the next `.array` is a new buffer at a
#     new address: a miss.  (The services' OWN closing `clear_cache()` calls, zncc_service.py:179-180, keep an in-memory raster's entry:
#     `keep_shared_across` below - nothing was edited in between.)
_SHARED: "dict[tuple, _SharedEntry]" = {}
_SHARED_LIMIT = 2
shared_pair_uploads = 0           # uploads `shared_pair` had to make (tests, diagnostics)


def _layout(arr: np.ndarray) -> tuple:
    a = np.asarray(arr)
    return (a.__array_interface__["data"][0], a.shape, a.strides, a.dtype.str)


def _same_buffer(a: np.ndarray, b: np.ndarray) -> bool:
    """Cheap necessary condition for two arrays to share memory: their byte ranges overlap."""
    (a0, a1), (b0, b1) = np.byte_bounds(a) if hasattr(np, "byte_bounds") else np.lib.array_utils.byte_bounds(a), \
        np.byte_bounds(b) if hasattr(np, "byte_bounds") else np.lib.array_utils.byte_bounds(b)
    return a0 < b1 and b0 < a1


def _base_chain(arr: np.ndarray) -> list:
    """`arr` and every ndarray it is a view of, outermost owner last."""
    out, a = [], arr
    while isinstance(a, np.ndarray):
        out.append(a)
        a = a.base
    return out


class _SharedEntry:
    def __init__(self, pair, mon, ref, rasters):
        self.pair, self.mon, self.ref = pair, mon, ref             # the host arrays stay referenced: their addresses cannot be recycled
        self.raster_ids = tuple(id(r) for r in rasters if r is not None)
        self.rasters = tuple(r for r in rasters if r is not None)  # (kept alive too: an id must not be reused by another raster)
        self.guarded = []
        seen = set()
        # the arrays that were uploaded, the arrays they are views of, and the array objects the rasters hand out as `.array` when
        # they are cached views of the same memory (`NumpyRasterImage._array`; a raster's cache is looked at, never loaded)
        cached = [a for a in (getattr(r, "_array", None) for r in self.rasters) if isinstance(a, np.ndarray)
                  and any(np.shares_memory(a, b, max_work=1) for b in (mon, ref) if _same_buffer(a, b))]
        chains = _base_chain(mon)[::-1] + _base_chain(ref)[::-1]
        for a in cached:
            chains += _base_chain(a)[::-1]
        for a in chains:  # owners first
            if id(a) not in seen and a.flags.writeable:
                seen.add(id(a))
                try:
                    a.flags.writeable = False
                    self.guarded.append(a)
                except ValueError:  # pragma: no cover - an exotic buffer that refuses
                    pass

    def intact(self, mon, ref) -> bool:
        """The lookup comes through arrays nobody could have written to since the upload."""
        return not any(a.flags.writeable for a in _base_chain(np.asarray(mon)) + _base_chain(np.asarray(ref)) + self.guarded)

    def release(self):
        for a in self.guarded:                                     # owners first: a view cannot become writeable before its base
            try:
                a.flags.writeable = True
            except ValueError:  # pragma: no cover
                pass
        self.guarded = []


def _drop(key) -> None:
    ent = _SHARED.pop(key, None)
    if ent is not None:
        ent.release()


def shared_pair(mon: np.ndarray, ref: np.ndarray, ctx: Context | None = None, publish: "ResidentPair | None" = None, rasters=()) -> "ResidentPair":
    """The resident copy of (mon, ref) on `ctx`: a pair published earlier for the same two host buffers (e.g. by `KLT.match` for
    a tile that covers the whole image) and still valid (see the token rules above), else a fresh upload that is remembered for the
    next service.  `publish` registers an existing pair instead of looking one up; `rasters`: the raster objects the arrays came
    from (their `clear_cache()` invalidates the entry).

    An upload from page-locked memory is asynchronous: the source must stay untouched until the pair's first device call
    (`ResidentPair._ready`) or `Context.sync()` - the read-only guard covers that window too."""
    global shared_pair_uploads
    ctx = ctx if ctx is not None else default_context()
    mon, ref = np.asarray(mon), np.asarray(ref)
    key = (id(ctx), _layout(mon), _layout(ref))
    if publish is None:
        hit = _SHARED.get(key)
        if hit is not None and hit.intact(mon, ref):
            return hit.pair
        _drop(key)
        publish = ResidentPair.upload(mon, ref, ctx=ctx)
        shared_pair_uploads += 1
    _drop(key)
    while len(_SHARED) >= _SHARED_LIMIT:
        _drop(next(iter(_SHARED)))
    _SHARED[key] = _SharedEntry(publish, mon, ref, rasters)
    return publish


def invalidate_raster(raster) -> None:
    """`raster.clear_cache()` was called by its owner: every shared pair made from it is dropped (write access comes back)."""
    for key in [k for k, e in _SHARED.items() if id(raster) in e.raster_ids]:
        _drop(key)


class keep_shared_across:
    """Context manager for the services' own closing `clear_cache()` calls (zncc_service.py:179-180): an in-memory raster keeps its
    pixels there, so entries whose host buffers are still the raster's current `.array` survive the call."""

    def __init__(self, *rasters):
        self.rasters = rasters

    def __enter__(self):
        self.saved = {k: e for k, e in _SHARED.items() if any(id(r) in e.raster_ids for r in self.rasters)}
        for k in self.saved:
            _SHARED.pop(k)                                        # (out of reach of invalidate_raster; still guarded)
        return self

    def __exit__(self, *exc):
        for k, e in self.saved.items():
            held = {_layout(e.mon)[0], _layout(e.ref)[0]}
            # survives iff every raster of the entry is an in-memory one whose current array is still the buffer that was uploaded
            keeps = all(getattr(r, "_keeps_array_across_clear_cache", False) and _layout(r.array)[0] in held for r in e.rasters)
            if keeps and len(_SHARED) < _SHARED_LIMIT:
                _SHARED[k] = e
            else:
                e.release()
        return False


def forget_shared_pairs() -> None:
    """Drop the shared resident pairs (their device buffers return to the context's pool, their host arrays become writeable again)."""
    for key in list(_SHARED):
        _drop(key)


__all__ = ["ResidentPair", "DeviceBuffer", "PendingBatch", "submit_units", "shared_pair", "forget_shared_pairs", "invalidate_raster", "_lib"]
