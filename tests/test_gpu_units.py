"""Batched units (km_klt_units_frame_submit, csrc/api_units.hip): U independent tiles (reference karios/matcher/klt.py:220-253: no halo,
per-tile stretch / threshold / maxCorners) through ONE set of device launches.  The contract is bit-identity with the unit-by-unit
submission - every kernel runs the single-unit item code on the unit's own rasters, scalar block, key buffer and grid - including the
score columns, units of different shape, units of different pairs, flagged units and the frame sink."""
import numpy as np
import pytest

from karios_amd import synth

pytestmark = pytest.mark.gpu


def same_rows(a, b) -> bool:
    """Two frame blocks hold the same frame: header and, column by column, the first n_rows entries (what lies behind them in a
    column is whatever an earlier, longer frame left there)."""
    ia, ib = a.block.view(np.int32), b.block.view(np.int32)
    if not np.array_equal(ia[:4], ib[:4]) or a.cap != b.cap or a.with_zncc != b.with_zncc:
        return False
    n, cap = int(ia[0]), a.cap
    cols32 = all(np.array_equal(ia[4 + k * cap:4 + k * cap + n], ib[4 + k * cap:4 + k * cap + n]) for k in range(6))
    base = 4 + 6 * cap
    cols64 = all(np.array_equal(ia[base + 2 * k * cap:base + 2 * k * cap + 2 * n], ib[base + 2 * k * cap:base + 2 * k * cap + 2 * n]) for k in range(int(a.with_zncc)))
    return cols32 and cols64


def _pairs(ops):
    from karios_amd.resident import ResidentPair
    ctx = ops._lib.default_context()
    mon_a, ref_a = synth.make_pair(1400, 1500, 0.5, 0.25, seed=11, nodata_wedge=True)
    mon_b, ref_b = synth.make_pair(900, 1100, -0.3, 0.4, seed=12)
    return ctx, ResidentPair.upload(mon_a, ref_a, ctx=ctx), ResidentPair.upload(mon_b, ref_b, ctx=ctx), (mon_a, ref_a)


UNITS_A = [(0, 0, 700, 600), (700, 0, 800, 600), (0, 600, 1500, 800), (300, 200, 640, 512), None]


@pytest.mark.parametrize("scores", [None, "zncc", "full"])
def test_units_submitted_together_equal_units_submitted_one_by_one(ops, O, scores):
    from karios_amd.core import KLTConfiguration
    from karios_amd.resident import submit_units
    ctx, pa, pb, _ = _pairs(ops)
    conf = KLTConfiguration(maxCorners=1200)
    thr = None if scores is None else 0.4
    mi = scores == "full"
    units = [(pa, b, None) for b in UNITS_A] + [(pb, None, None), (pb, (100, 50, 900, 700), (5100, 7050))]
    pend = [p.submit_tile(conf, box=b, zncc_threshold=thr, origin=o, mutual_info=mi) for p, b, o in units[:3]]
    first = [q.wait() for q in pend]
    first += [p.submit_tile(conf, box=b, zncc_threshold=thr, origin=o, mutual_info=mi).wait() for p, b, o in units[3:]]
    want = [p.submit_tile(conf, box=b, zncc_threshold=thr, origin=o, mutual_info=mi).result() for p, b, o in units]
    assert all(w.n_rows > 200 and w.flags == 0 for w in want)
    assert sum(1 for f in first if f.flags == 0) >= 4          # (a unit may exceed the fixed capacities by itself: then the batch flags it too)
    for rep in range(3):                                   # (slot ring, workspace reuse, the early min / max behind another submission)
        batch = submit_units(units, conf, thr, mi)
        assert batch is not None and len(batch) == len(units)
        got = list(batch.wait())
        for k, (g, f, w) in enumerate(zip(got, first, want)):
            # bit for bit what the unit's own submission delivered - flags of the synchronisation-free corner path included (the ROWS of
            # a flagged block are not a result: only its header is compared)
            assert g.flags == f.flags, (rep, k, g.flags, f.flags)
            if g.flags:
                assert np.array_equal(g.block[:4].view(np.int32), f.block[:4].view(np.int32)), (rep, k)
                got[k] = g = batch.redo(k)
            else:
                assert same_rows(g, f), (rep, k)
            assert g.flags == 0 and same_rows(g, w), (rep, k)
    # one unit against the oracle: the batch is not merely self-consistent
    exp = O.klt_tile(*_pairs_host(3), O.default_conf(maxCorners=1200))
    f = got[3].to_frame()
    np.testing.assert_array_equal(f["x0"].to_numpy(), exp["x0"] + 300)
    np.testing.assert_array_equal(f["y0"].to_numpy(), exp["y0"] + 200)
    np.testing.assert_array_equal(f["dx"].to_numpy(), exp["dx"])


def _pairs_host(k):
    mon_a, ref_a = synth.make_pair(1400, 1500, 0.5, 0.25, seed=11, nodata_wedge=True)
    x, y, w, h = UNITS_A[k]
    return mon_a[y:y + h, x:x + w], ref_a[y:y + h, x:x + w]


def test_a_flagged_unit_of_a_batch_is_repeated_alone_and_the_sink_receives_every_block(ops):
    import ctypes as C
    from karios_amd.core import KLTConfiguration
    from karios_amd.resident import DeviceBuffer
    from karios_amd.stream import FrameStream
    from karios_amd import frames
    ctx, pa, pb, _ = _pairs(ops)
    conf = KLTConfiguration(maxCorners=900)
    units = [(pa, UNITS_A[1], None), (pa, UNITS_A[2], None), (pb, None, None), (pb, (100, 50, 900, 700), None)]
    with FrameStream(0.4, depth=1) as s:
        want = []
        for p, b, o in units:
            want += s.submit(p, conf, b, o)
        want += s.drain()
    words = frames.block_words(conf.maxCorners, 1)
    pitch = 4 * words + 64
    sink = DeviceBuffer(ctx, pitch * len(units))
    ctx.set_frame_sink(sink.ptr, pitch * len(units), pitch)
    try:
        with FrameStream(0.4, depth=1) as s:
            got = s.submit_many(units, conf, tags=list("abcd")) + s.drain()
            ctx.sync()
            blocks = sink.download((len(units), pitch // 4), np.float32)
            ctx.set_option("spec_flag", 32)
            flagged = s.submit_many(units, conf) + s.drain()
            ctx.set_option("spec_flag", 0)
    finally:
        ctx.set_option("spec_flag", 0)
        ctx.set_frame_sink(None)
    assert [d.tag for d in got] == list("abcd") and not any(d.redone for d in got) and not any(d.redone for d in want)
    for k, (g, w) in enumerate(zip(got, want)):
        assert same_rows(g.raw, w.raw), k
        from karios_amd.resident import RawFrame
        assert same_rows(RawFrame(blocks[k, :words], conf.maxCorners, 1), w.raw), k                   # the sink holds unit k's block at k * pitch
        assert g.frame.equals(w.frame)
    assert all(d.redone and d.flags & 32 for d in flagged)
    for k, (g, w) in enumerate(zip(flagged, want)):
        assert g.raw.flags == 0 and g.frame.equals(w.frame), k


def test_klt_match_on_resident_rasters_goes_through_batched_tiles(ops, O):
    """`KLT.match` (klt.py:198-234) on rasters that live in HBM: the tile grid - unequal edge tiles included - is one batched submission;
    frames in x-outer / y-inner order, each equal to the oracle's tile."""
    import torch
    from karios_amd.core import KLTConfiguration
    from karios_amd.core.image import DeviceRasterImage
    from karios_amd.matcher import KLT
    dev = torch.device("cuda", 0)
    mon_t, ref_t = synth.make_pair_torch(1300, 1700, 0.4, -0.2, seed=9, device=dev)
    torch.cuda.synchronize()
    conf = KLTConfiguration(tile_size=900, maxCorners=800, laplacian_kernel_size=5)
    klt = KLT(conf)
    frames_ = list(klt.match(DeviceRasterImage(mon_t, np.uint16), DeviceRasterImage(ref_t, np.uint16), None))
    grid = klt.tile_boxes(1700, 1300)
    assert [tuple(t) for t in grid] == [(0, 0, 900, 900), (0, 900, 900, 400), (900, 0, 800, 900), (900, 900, 800, 400)] and len(frames_) == 4
    mon, ref = mon_t.cpu().numpy().view(np.uint16), ref_t.cpu().numpy().view(np.uint16)
    oc = O.default_conf(maxCorners=800, laplacian_kernel_size=5, tile_size=900)
    for t, f in zip(grid, frames_):
        exp = O.klt_tile(mon[t.y_off:t.y_off + t.y_size, t.x_off:t.x_off + t.x_size], ref[t.y_off:t.y_off + t.y_size, t.x_off:t.x_off + t.x_size], oc,
                         x_off=t.x_off, y_off=t.y_off)
        for col in ("x0", "y0", "dx", "dy", "score"):
            np.testing.assert_array_equal(f[col].to_numpy(), exp[col], err_msg=f"{tuple(t)} {col}")
        assert list(f.columns) == ["x0", "y0", "dx", "dy", "score"] and f["x0"].dtype == np.float32


def test_what_the_batch_form_does_not_cover_goes_one_by_one(ops):
    from karios_amd.core import KLTConfiguration
    from karios_amd.resident import submit_units
    from karios_amd.stream import FrameStream
    ctx, pa, pb, _ = _pairs(ops)
    narrow = [(pa, (0, 0, 400, 600), None), (pa, (400, 0, 500, 600), None)]          # narrower than 512 columns
    assert submit_units(narrow, KLTConfiguration(maxCorners=500)) is None
    assert submit_units([(pa, None, None), (pb, None, None)], KLTConfiguration(maxCorners=0)) is None
    # kernel sizes 9 and 11 ARE covered since round 6 (9: the marching Laplacian pass holds nine taps; 11: the LDS kernel, a launch per
    # unit inside the batch): the batch equals the units one by one
    for ks in ({"mon": 9, "ref": 7}, 11, {"mon": 9, "ref": 11}):
        confk = KLTConfiguration(maxCorners=500, laplacian_kernel_size=ks)
        got = submit_units([(pa, None, None), (pb, None, None)], confk, 0.4)
        assert got is not None, ks
        for k, (g, p) in enumerate(zip(got.wait(), (pa, pb))):
            w = p.submit_tile(confk, zncc_threshold=0.4).result()
            g = got.redo(k) if g.flags else g
            assert w.n_rows > 100 and same_rows(g, w), (ks, k)
    conf = KLTConfiguration(maxCorners=500)
    with FrameStream(None, depth=1) as s:
        a = s.submit_many(narrow, conf) + s.drain()
    with FrameStream(None, depth=1) as s:
        b = []
        for p, bx, o in narrow:
            b += s.submit(p, conf, bx, o)
        b += s.drain()
    assert len(a) == 2 and all(x.frame.equals(y.frame) for x, y in zip(a, b))


# ---------------------------------------------------------------------------- software pipeline over consecutive submissions (round 6)
def test_pipelined_submissions_equal_unpipelined_ones_bit_for_bit(ops):
    """km_set_option("units_pipeline", 1): two workspace lanes alternate, the dense stages of consecutive submissions interleave and the
    latency-bound chains of one run beside the dense kernels of the next; the tail of a submission is enqueued by the next one, by a
    flush, by any other entry point or by the context's sync.  Same kernels on the same data: every frame block is the unpipelined
    submission's bit for bit - across alternating batches of different shape, with all score columns, through every way a tail can be
    enqueued."""
    import threading
    from karios_amd.core import KLTConfiguration
    from karios_amd.resident import submit_units
    ctx, pa, pb, _ = _pairs(ops)
    conf = KLTConfiguration(maxCorners=1200)
    batch_x = [(pa, b, None) for b in UNITS_A[:3]] + [(pb, None, None)]
    batch_y = [(pb, (100, 50, 900, 700), (5100, 7050)), (pa, UNITS_A[3], None), (pa, None, None)]
    want = {}
    for name, units in (("x", batch_x), ("y", batch_y)):
        b = submit_units(units, conf, 0.4, True)
        want[name] = [r if not r.flags else b.redo(k) for k, r in enumerate(b.wait())]
    assert all(w.n_rows > 200 for v in want.values() for w in v)

    def check(batch, name, phase=""):
        got = batch.wait()
        for k, (g, w) in enumerate(zip(got, want[name])):
            if g.flags:
                g = batch.redo(k)
            assert same_rows(g, w), (phase, name, k, g.block[:4].view(np.int32).tolist(), w.block[:4].view(np.int32).tolist())

    ctx.set_option("units_pipeline", 1)
    try:
        # 1. a stream: submission i is waited for after submission i + 1 went in (its tail travelled with that one)
        seq = ["x", "y", "x", "x", "y", "y", "x"]
        prev = None
        for name in seq:
            cur = (submit_units(batch_x if name == "x" else batch_y, conf, 0.4, True), name)
            if prev is not None:
                check(*prev, "stream")
            prev = cur
        check(*prev, "stream-last")                                      # the last one: PendingBatch.wait() on the submitting thread flushes its tail
        # 2. waited for at once, every time (flush per submission: the pipeline degenerates, the frames do not)
        for name in ("y", "x", "y"):
            check(submit_units(batch_x if name == "x" else batch_y, conf, 0.4, True), name, "at-once")
        # 3. another entry point between two submissions enqueues the deferred tail first (stream order is kept)
        a = submit_units(batch_x, conf, 0.4, True)
        single = pa.match_tile_raw(conf, UNITS_A[3], 0.4, mutual_info=True)
        b = submit_units(batch_y, conf, 0.4, True)
        ctx.sync()                                        # ... and so does the context's sync
        check(a, "x", "entry-point")
        check(b, "y", "entry-point")
        assert same_rows(single, want["y"][1])
        # 4. a worker thread waits while the submitting thread goes on: the next submission releases it
        a = submit_units(batch_x, conf, 0.4, True)
        out = {}
        t = threading.Thread(target=lambda: out.setdefault("raws", a.wait()))
        t.start()
        b = submit_units(batch_y, conf, 0.4, True)
        t.join(timeout=30)
        assert not t.is_alive() and len(out["raws"]) == len(batch_x)
        check(a, "x")
        check(b, "y")
        # 5. nobody flushes: km_frame_wait enqueues the tail itself after 100 ms instead of waiting for ever
        a = submit_units(batch_y, conf, 0.4, True)
        t = threading.Thread(target=lambda: out.setdefault("late", a.wait()))
        t.start()
        t.join(timeout=30)
        assert not t.is_alive() and len(out["late"]) == len(batch_y)
        check(a, "y")
    finally:
        ctx.set_option("units_pipeline", 0)
    check(submit_units(batch_x, conf, 0.4, True), "x")     # and back


def test_framestream_pipelines_batched_submissions_and_yields_the_same_frames(ops):
    """FrameStream switches the pipeline on for the contexts it drives (depth >= 1) and flushes before it blocks on the newest
    submission: frames equal the unpipelined stream's, the option is restored when the stream closes."""
    from karios_amd.core import KLTConfiguration
    from karios_amd.stream import FrameStream
    ctx, pa, pb, _ = _pairs(ops)
    conf = KLTConfiguration(maxCorners=1200)
    units = [(pa, b, None) for b in UNITS_A[:4]] + [(pb, None, None)]
    runs = {}
    for piped in (False, True):
        frames = []
        with FrameStream(0.4, depth=2, pipeline=piped) as s:
            for _ in range(5):
                frames += [d.frame for d in s.submit_many(units, conf)]
                assert ctx.get_option("units_pipeline", 0) == int(piped)
            frames += [d.frame for d in s.drain()]
        assert ctx.get_option("units_pipeline", 0) == 0
        runs[piped] = frames
    assert len(runs[True]) == len(runs[False]) == 25
    for a, b in zip(runs[True], runs[False]):
        assert list(a.columns) == list(b.columns) and len(a) == len(b) > 200
        for col in a.columns:
            np.testing.assert_array_equal(a[col].to_numpy(), b[col].to_numpy())
    # every second submission of a pipelined stream flagged ("spec_flag"): its units are repeated exactly while the neighbours' tails are
    # still deferred / in flight on the other lane - the repeat (a blocking entry point) enqueues them first; frames unchanged
    frames, redone = [], []
    with FrameStream(0.4, depth=2) as s:
        for k in range(6):
            ctx.set_option("spec_flag", 32 if k % 2 else 0)
            done = s.submit_many(units, conf)
            ctx.set_option("spec_flag", 0)
            frames += [d.frame for d in done]
            redone += [d.redone for d in done]
        done = s.drain()
        frames += [d.frame for d in done]
        redone += [d.redone for d in done]
    assert len(frames) == 30 and all(redone[10 * k + 5:10 * k + 10] == [True] * 5 for k in range(3))    # (an unflagged unit may still exceed the fixed capacities by itself)
    for a, b in zip(frames, runs[False] + runs[False][:5]):
        for col in a.columns:
            np.testing.assert_array_equal(a[col].to_numpy(), b[col].to_numpy())


# ---------------------------------------------------------------------------- user masks in a batch (klt.py:258-266)
def test_units_with_a_user_mask_equal_the_masked_tiles_one_by_one_and_the_oracle(ops, O):
    """A pair with a user mask (the caller's uint8 raster instead of the automatic mask) tiled into a batched submission: every unit's
    box of the mask is packed and counted inside the batch's Laplacian stage.  Frames = the masked tile calls' bit for bit, pipelined or
    not, and one tile against the oracle with the same mask box; units that mix masked and unmasked pairs go one by one (None)."""
    from karios_amd.core import KLTConfiguration
    from karios_amd.resident import ResidentPair, submit_units
    from karios_amd.stream import FrameStream
    ctx = ops._lib.default_context()
    mon, ref = synth.make_pair(1300, 1480, 0.4, -0.3, seed=21)
    rng = np.random.default_rng(5)
    mask = np.ones(mon.shape, np.uint8)
    for _ in range(14):                                           # seeded rectangles zeroing ~ a fifth (BASELINE config 5's mask)
        y, x = int(rng.integers(0, 1200)), int(rng.integers(0, 1380))
        mask[y:y + int(rng.integers(40, 260)), x:x + int(rng.integers(40, 300))] = 0
    mask[::7, ::5] *= 3                                            # any non-zero byte is "valid" (klt.py:266 reads the raster as is)
    pair = ResidentPair.upload(mon, ref, mask, ctx=ctx)
    plain = ResidentPair.upload(mon, ref, ctx=ctx)
    conf = KLTConfiguration(maxCorners=1500)
    boxes = [(0, 0, 740, 650), (740, 0, 740, 650), (0, 650, 740, 650), (740, 650, 740, 650), (101, 77, 1011, 901)]
    units = [(pair, b, None) for b in boxes]
    want = [pair.submit_tile(conf, box=b, zncc_threshold=0.4).result() for b in boxes]
    assert all(w.n_rows > 300 and w.flags == 0 for w in want)
    unmasked = plain.submit_tile(conf, box=boxes[0], zncc_threshold=0.4).result()
    assert not same_rows(unmasked, want[0])                       # the mask matters on this content
    for piped in (0, 1):
        ctx.set_option("units_pipeline", piped)
        try:
            pend = [submit_units(units, conf, 0.4) for _ in range(3)]
            assert all(p is not None for p in pend)
            for p in pend:
                for k, (g, w) in enumerate(zip(p.wait(), want)):
                    g = p.redo(k) if g.flags else g
                    assert same_rows(g, w), (piped, k)
        finally:
            ctx.set_option("units_pipeline", 0)
    conf11 = KLTConfiguration(maxCorners=1500, laplacian_kernel_size={"mon": 11, "ref": 5})   # (kernel 11: the batch's LDS-kernel form)
    p11 = submit_units(units[:3], conf11, 0.4)
    assert p11 is not None
    for k, g in enumerate(p11.wait()):
        w = pair.submit_tile(conf11, box=boxes[k], zncc_threshold=0.4).result()
        assert w.n_rows > 200 and same_rows(p11.redo(k) if g.flags else g, w), k
    assert submit_units(units[:2] + [(plain, boxes[2], None)], conf, 0.4) is None          # masked and unmasked units do not mix
    with FrameStream(0.4, depth=1) as s:                          # ... FrameStream then submits those one by one, in order
        mixed = s.submit_many(units[:2] + [(plain, boxes[2], None)], conf) + s.drain()
    assert len(mixed) == 3 and same_rows(mixed[0].raw, want[0]) and same_rows(mixed[1].raw, want[1])
    x, y, w, h = boxes[4]
    exp = O.klt_tile(mon[y:y + h, x:x + w], ref[y:y + h, x:x + w], O.default_conf(maxCorners=1500), mask_box=mask[y:y + h, x:x + w])
    f = want[4].to_frame()
    np.testing.assert_array_equal(f["x0"].to_numpy(), exp["x0"] + x)
    np.testing.assert_array_equal(f["y0"].to_numpy(), exp["y0"] + y)
    np.testing.assert_array_equal(f["dx"].to_numpy(), exp["dx"])
