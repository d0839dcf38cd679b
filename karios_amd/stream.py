"""Pipelined matching of a stream of tiles / band pairs: the device half of tile i+1 runs while a worker thread turns the
finished block of tile i into its DataFrame.

`KariosAPI._compute_matches` (reference `karios/api/core.py:845-921`) consumes `KLT.match` one frame at a time and post-processes
each (`_handle_klt_results`: radial error, angle, ZNCC of the confident rows) before asking for the next; the device would idle
during that host work.  `FrameStream` is the product-side form of that loop: the caller SUBMITS work units (any `ResidentPair`,
any box, on any number of library contexts) and receives finished, scored frames in submission order.  Everything timing-
related lives here - the bounded number of units in flight, the worker thread, the interpreter's switch interval while the
stream is open, the exact repeat of a unit the synchronisation-free corner path flagged - so that `bench.py`, `KLT.match`
callers and `ResidentPair.match_pipelined` share one implementation.
"""
from __future__ import annotations

import sys
import threading
from collections import deque
from concurrent.futures import ThreadPoolExecutor
from dataclasses import dataclass, field

from pandas import DataFrame

from .resident import PendingBatch, PendingFrame, RawFrame, ResidentPair, submit_units


# The interpreter's switch interval is process-global: open streams are counted, the original value is saved by the FIRST one
# and restored by the LAST one to close (two overlapping streams closing out of order would otherwise leave it lowered for good).
_gil_lock = threading.Lock()
_gil_open = 0
_gil_saved = None


def _gil_interval_acquire(interval: float) -> None:
    global _gil_open, _gil_saved
    with _gil_lock:
        if _gil_open == 0:
            _gil_saved = sys.getswitchinterval()
        _gil_open += 1
        if sys.getswitchinterval() > interval:
            sys.setswitchinterval(interval)


def _gil_interval_release() -> None:
    global _gil_open, _gil_saved
    with _gil_lock:
        _gil_open -= 1
        if _gil_open == 0 and _gil_saved is not None:
            sys.setswitchinterval(_gil_saved)
            _gil_saved = None


@dataclass
class StreamResult:
    """One finished work unit."""
    frame: DataFrame | None      # None: no valid pixel / no corner in the unit (the reference yields nothing for such a tile)
    raw: RawFrame                # the unit's frame block (the unit of the multi-GPU gather)
    tag: object = None           # whatever the caller passed to `submit`
    redone: bool = False         # the unit was repeated through the exact corner path
    spans: dict = field(default_factory=dict)   # stage -> ms (only with `Context.set_profiling(True)` and `want_spans`)
    flags: int = 0               # KM_FLAG_* bits the synchronisation-free corner path raised for the unit (0: it was not repeated)


class FrameStream:
    """`with FrameStream(confidence_threshold=0.4) as s: for ...: done += s.submit(pair, conf)`; `done += s.drain()`.

    depth               units still pending when `submit` returns (1: classic two-stage pipeline - the host half of unit i runs
                        beside the device half of unit i+1; 0: synchronous; 2 x contexts when the pairs sit on several contexts
                        of one GPU - a context holds at most three frames in flight)
    confidence_threshold  rows with score >= threshold get their ZNCC in the SAME device call (`zncc_score` column); None: bare frames
    mutual_info         with a threshold: the SAME device call also scores `mutual_info_score` and `mi_score` of those rows - the
                        whole of `_handle_klt_results`' scoring (core.py:894-907) behind the tile, no second pass over the key points
    score_columns       the frame also carries the `radial error` / `angle` columns of `_handle_klt_results` (core.py:872-893);
                        default: whenever a threshold is given
    host_stage          optional callable(frame, pair) -> frame run on the CALLING thread when a frame is collected (a stage that
                        calls back into the library - mutual information, ZNCC of rows the device did not score - must not run
                        beside `submit`: a context is not thread-safe)
    pipeline            batched submissions (`submit_many`) run as a SOFTWARE PIPELINE on their context (km_set_option "units_pipeline",
                        csrc/api_units.hip): the dense stages of consecutive submissions interleave, the latency-bound chains of one run
                        beside the dense kernels of the next; a submission's tail is enqueued by the next one - the stream flushes it
                        (`Context.flush`) before it blocks on the newest submission and in `drain`.  Needs depth >= 1; frames are
                        bit-identical.  Default: on.
    gil_switch_interval the submitting thread spends ~0.1 ms per unit inside the library and the rest in Python next to the worker;
                        with CPython's default 5 ms switch interval a thread that needs the GIL can wait that long for the other to
                        yield it.  The stream lowers the (process-global!) interval while it is open; the value found by the first
                        open stream comes back when the last one closes (reference-counted).  None leaves it alone.
    """

    def __init__(self, confidence_threshold: float | None = None, depth: int = 1, host_stage=None, want_spans: bool = False,
                 gil_switch_interval: float | None = 1e-4, score_columns: bool | None = None, mutual_info: bool = False, pipeline: bool = True):
        self.threshold, self.depth, self.host_stage, self.want_spans = confidence_threshold, max(0, int(depth)), host_stage, want_spans
        self.mutual_info = bool(mutual_info) and confidence_threshold is not None
        self.score_columns = (confidence_threshold is not None) if score_columns is None else bool(score_columns)
        self._pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="karios-frame")
        self._pending: deque = deque()
        self._holds_interval = False
        if gil_switch_interval is not None:
            _gil_interval_acquire(gil_switch_interval)
            self._holds_interval = True
        self.pipeline = bool(pipeline) and self.depth >= 1
        self._piped = {}               # contexts whose "units_pipeline" option this stream switched on -> (context, previous value)
        self.units_redone = 0
        self.worker_cpu_s = 0.0        # CPU time of the worker thread's host halves (time.thread_time): what a rank's frames cost its host

    # ------------------------------------------------------------------ context manager
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def close(self):
        for ctx, before in self._piped.values():
            try:
                if getattr(ctx, "handle", None):
                    ctx.set_option("units_pipeline", before)      # (0 flushes what is still deferred)
            except Exception:  # pragma: no cover - a context that was closed under the stream
                pass
        self._piped = {}
        if self._pool is not None:
            self._pool.shutdown(wait=True)
            self._pool = None
        if self._holds_interval:
            self._holds_interval = False
            _gil_interval_release()

    def __del__(self):
        try:
            self.close()           # an abandoned stream (a generator that was never exhausted) must not keep the interval lowered
        except Exception:  # pragma: no cover - interpreter shutdown
            pass

    # ------------------------------------------------------------------ the two halves
    def _host_half(self, pair: ResidentPair, pend):
        """Worker thread: touches the frame slot (km_frame_wait) and numpy / pandas only."""
        import time
        t_cpu = time.thread_time()
        try:
            return self._host_half_body(pair, pend)
        finally:
            self.worker_cpu_s += time.thread_time() - t_cpu

    def _host_half_body(self, pair: ResidentPair, pend):
        raw = pend.wait() if isinstance(pend, PendingFrame) else pend
        spans = pend.stage_ms() if self.want_spans and isinstance(pend, PendingFrame) else {}
        frame = None
        if not raw.flags:
            frame = raw.to_frame(radial=self.score_columns)
            if frame is not None and self.score_columns and raw.with_zncc:
                frame = pair.score_frame(frame, self.threshold, mutual_info=self.mutual_info)     # every column is in place: pure numpy, no library call
        return raw, spans, frame

    def _batch_host_half(self, pairs, pend: PendingBatch):
        """Worker thread, batched submission: the units' blocks arrive together."""
        import time
        t_cpu = time.thread_time()
        try:
            return self._batch_host_half_body(pairs, pend)
        finally:
            self.worker_cpu_s += time.thread_time() - t_cpu

    def _batch_host_half_body(self, pairs, pend: PendingBatch):
        raws = pend.wait()
        spans = pend.stage_ms() if self.want_spans else {}
        out = []
        for pair, raw in zip(pairs, raws):
            frame = None
            if not raw.flags:
                frame = raw.to_frame(radial=self.score_columns)
                if frame is not None and self.score_columns and raw.with_zncc:
                    frame = pair.score_frame(frame, self.threshold, mutual_info=self.mutual_info)
            out.append((raw, frame))
        return out, spans

    def _collect_batch(self, item) -> list[StreamResult]:
        pairs, pend, tags, fut = item
        done, spans = fut.result()
        out = []
        for i, (pair, (raw, frame), tag) in enumerate(zip(pairs, done, tags)):
            redone, flags = False, raw.flags
            if raw.flags:                  # this unit alone through the exact path (the others of the batch stand)
                raw = pend.redo(i)
                frame = raw.to_frame(radial=self.score_columns)
                if frame is not None and self.score_columns and self.threshold is not None:
                    frame = pair.score_frame(frame, self.threshold, mutual_info=self.mutual_info)
                redone = True
                self.units_redone += 1
            if frame is not None and self.host_stage is not None:
                frame = self.host_stage(frame, pair)
            # the batch's stage spans are attached to its FIRST unit (they cover all units: one set of launches)
            out.append(StreamResult(frame, raw, tag, redone, spans if i == 0 else {}, flags))
        return out

    def _collect(self, item):
        if isinstance(item[1], PendingBatch):
            # the NEWEST submission of a pipelined context still lacks its tail: enqueue it before blocking on its frame
            if self._piped:
                item[1].ctx.flush(item[1].ticket)
            return self._collect_batch(item)
        return [self._collect_one(item)]

    def _collect_one(self, item) -> StreamResult:
        pair, pend, tag, fut = item
        raw, spans, frame = fut.result()
        redone, flags = False, raw.flags
        if raw.flags:                      # did not fit the fixed capacities of the synchronisation-free corner path: exact repeat, here
            raw = pend.redo()
            frame = raw.to_frame(radial=self.score_columns)
            if frame is not None and self.score_columns and self.threshold is not None:
                frame = pair.score_frame(frame, self.threshold, mutual_info=self.mutual_info)
            redone = True
            self.units_redone += 1
        if frame is not None and self.host_stage is not None:
            frame = self.host_stage(frame, pair)
        return StreamResult(frame, raw, tag, redone, spans, flags)

    # ------------------------------------------------------------------ API
    def submit(self, pair: ResidentPair, conf, box=None, origin=None, tag=None, on_submitted=None) -> list[StreamResult]:
        """Queue one unit on `pair`'s context; returns the units collected to keep at most `depth` pending (oldest first -
        possibly none).  `on_submitted(pending_frame)` runs on the calling thread right behind the submission (e.g. the hand-over
        of the unit's block to a collective: `karios_amd.parallel.RankBlockExchange.issue`)."""
        if self._pool is None:
            raise RuntimeError("FrameStream is closed")
        # maxCorners == 0 (unbounded) sizes the frame block for a quarter of the tile's pixels: no pinned 3-slot ring for that
        if conf.maxCorners > 0:
            pend = pair.submit_tile(conf, box, self.threshold, origin, mutual_info=self.mutual_info)
        else:
            pend = pair.match_tile_raw(conf, box, self.threshold, origin, mutual_info=self.mutual_info)    # (blocking; repeats a flagged tile itself)
        if on_submitted is not None:
            on_submitted(pend)
        self._pending.append((pair, pend, tag, self._pool.submit(self._host_half, pair, pend)))
        out = []
        while len(self._pending) > self.depth:
            out += self._collect(self._pending.popleft())
        return out

    def submit_many(self, units, conf, tags=None, on_submitted=None) -> list[StreamResult]:
        """Queue several independent units - `units` = [(pair, box | None, origin | None), ...], e.g. the tiles of one `KLT.match`
        (klt.py:220-253) or a rank's share of a multi-band job - as BATCHED submissions (`karios_amd.resident.submit_units`: one set
        of device launches per <= 16 units; a batch counts as one pending submission for `depth`).  Units the batch form does not
        cover (another context, a user mask, maxCorners 0 ...) go one by one through `submit`, in order.  Returns the collected
        results like `submit`; `on_submitted(pending_batch_or_frame, first_unit_index)` runs behind every submission."""
        from ._lib import UNITS_PER_SUBMISSION
        if self._pool is None:
            raise RuntimeError("FrameStream is closed")
        tags = list(tags) if tags is not None else [None] * len(units)
        out = []
        i = 0
        while i < len(units):
            chunk, chunk_tags = units[i:i + UNITS_PER_SUBMISSION], tags[i:i + UNITS_PER_SUBMISSION]
            if self.pipeline and len(chunk) > 1:
                ctx = chunk[0][0].ctx
                if id(ctx) not in self._piped:
                    self._piped[id(ctx)] = (ctx, ctx.get_option("units_pipeline", 0))
                    ctx.set_option("units_pipeline", 1)
            pend = submit_units(chunk, conf, self.threshold, self.mutual_info) if len(chunk) > 1 else None
            if pend is None:               # not batchable: unit by unit
                for (pair, box, origin), tag in zip(chunk, chunk_tags):
                    out += self.submit(pair, conf, box, origin, tag, None if on_submitted is None else (lambda p, k=i: on_submitted(p, k)))
                    i += 1
                continue
            if on_submitted is not None:
                on_submitted(pend, i)
            pairs = [u[0] for u in chunk]
            self._pending.append((pairs, pend, chunk_tags, self._pool.submit(self._batch_host_half, pairs, pend)))
            while len(self._pending) > self.depth:
                out += self._collect(self._pending.popleft())
            i += len(chunk)
        return out

    def drain(self) -> list[StreamResult]:
        # no further submission follows: the newest batch of every pipelined context gets its tail NOW (its LK then queues right behind
        # its eigenvalue pass instead of waiting until the older batches have been collected)
        if self._piped:
            newest = {}
            for item in self._pending:
                if isinstance(item[1], PendingBatch):
                    newest[id(item[1].ctx)] = item[1]
            for pend in newest.values():
                pend.ctx.flush(pend.ticket)
        out = []
        while self._pending:
            out += self._collect(self._pending.popleft())
        return out


__all__ = ["FrameStream", "StreamResult"]
