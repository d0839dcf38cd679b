"""Driver of the host-only sanitizer build (run by tests/test_host_asan.py in a subprocess with libasan preloaded).

Loads tests/hoststub/_build/libkarios_host_asan.so - api.hip + staging.hip compiled with g++ -fsanitize=address,undefined against
the stand-in HIP layer - through the SAME ctypes signatures the product uses (karios_amd._lib.SIGNATURES) and walks the host-side
bookkeeping: argument validation of every family of entry points, workspace slots (regrow, allocation failure), the page-locked
staging ring and landing arena (sizes around the chunk boundaries, strided sources), upload tickets, the three-slot frame ring.
Prints 'HOST-ASAN OK' at the end; any sanitizer report aborts the process.
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from karios_amd._lib import SIGNATURES, KltParams, KltStats, KmUnit  # noqa: E402  (signatures only: the product library is NOT loaded)

lib = C.CDLL(os.path.join(ROOT, "tests", "hoststub", "_build", "libkarios_host_asan.so"))
for name, (res, args) in SIGNATURES.items():
    fn = getattr(lib, name)
    fn.restype, fn.argtypes = res, args
lib.stub_counters.argtypes = [C.POINTER(C.c_long)]
lib.stub_fail_malloc_after.argtypes = [C.c_int]

KM_E_ARG, KM_E_NOMEM = -1, -3


def P(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def ok(rc, what):
    assert rc == 0, f"{what}: status {rc}: {lib.km_last_error(ctx).decode()}"


def err(rc, what, code=None):
    assert rc < 0 and (code is None or rc == code), f"{what}: expected an error, got {rc}"


def params(**kw):
    p = KltParams()
    p.max_corners, p.block_size, p.win_size, p.max_level, p.max_count = 50, 15, 25, 1, 30
    p.ksize_mon = p.ksize_ref = 7
    p.quality_level, p.min_distance, p.epsilon = 0.1, 10.0, 0.03
    for k, v in kw.items():
        setattr(p, k, v)
    return p


rng = np.random.default_rng(3)
ctx = C.c_void_p()
err(lib.km_ctx_create(0, None), "ctx_create(NULL)")
err(lib.km_ctx_create(7, C.byref(ctx)), "ctx_create(device 7)")
ok(lib.km_ctx_create(0, C.byref(ctx)), "ctx_create")
err(lib.km_set_option(ctx, b"no_such_option", 1), "unknown option", KM_E_ARG)
ok(lib.km_set_option(ctx, b"speculative", 1), "set_option")

# ---- staging ring / landing arena: round trips around every boundary (the ring is 4 x 64 KB in this run, the arena 8 MB)
chunk = int(os.environ["KARIOS_HIP_RING_CHUNK_KB"]) << 10
for n in (1, 63, 64, chunk - 1, chunk, chunk + 1, 4 * chunk, 4 * chunk + 17, 9 * chunk + 5, (4 << 20) + 3, (8 << 20) + 1, (21 << 20) + 7):
    d = C.c_void_p()
    ok(lib.km_dev_alloc(ctx, n, C.byref(d)), "dev_alloc")
    src = rng.integers(0, 256, n, dtype=np.uint8)
    back = np.zeros(n, np.uint8)
    ok(lib.km_h2d(ctx, d, P(src), n), "h2d")
    ok(lib.km_d2h(ctx, P(back), d, n), "d2h")
    assert np.array_equal(src, back), f"round trip of {n} bytes"
    ok(lib.km_dev_free(ctx, d), "dev_free")

# ---- strided pageable uploads on the copy stream, tickets
big = rng.integers(0, 60000, (300, 500)).astype(np.uint16)
view = big[13:290, 21:477]                      # 277 x 456, row stride 500
d = C.c_void_p()
ok(lib.km_dev_alloc(ctx, view.size * 2, C.byref(d)), "dev_alloc")
t1, t2 = C.c_int(-2), C.c_int(-2)
ok(lib.km_upload_mark(ctx, C.byref(t1)), "mark before any upload")
assert t1.value == -1
ok(lib.km_upload_async(ctx, d, view.shape[1] * 2, P(view), big.strides[0], view.shape[1] * 2, view.shape[0]), "upload_async (pageable, strided)")
err(lib.km_upload_async(ctx, d, 10, P(view), big.strides[0], view.shape[1] * 2, view.shape[0]), "upload_async pitch", KM_E_ARG)
ok(lib.km_upload_mark(ctx, C.byref(t1)), "mark")
ok(lib.km_upload_mark(ctx, C.byref(t2)), "mark 2")
assert t1.value >= 0 and t2.value >= 0 and t1.value != t2.value
ok(lib.km_upload_join(ctx, t1.value), "join")
err(lib.km_upload_join(ctx, t1.value), "join twice", KM_E_ARG)
err(lib.km_upload_join(ctx, 99), "join unknown", KM_E_ARG)
ok(lib.km_upload_join(ctx, -1), "join(-1)")
t3 = C.c_int()
ok(lib.km_upload_mark(ctx, C.byref(t3)), "mark 3")
assert t3.value == t1.value                     # the slot is reused
ok(lib.km_upload_join(ctx, t2.value), "join 2")
ok(lib.km_upload_join(ctx, t3.value), "join 3")
back = np.zeros(view.shape, np.uint16)
ok(lib.km_d2h(ctx, P(back), d, back.nbytes), "d2h")
assert np.array_equal(back, view)
ok(lib.km_dev_free(ctx, d), "dev_free")

# ---- page-locked source: DMA'd in place
h = C.c_void_p()
ok(lib.km_host_alloc(ctx, 1 << 16, C.byref(h)), "host_alloc")
pin = np.ctypeslib.as_array(C.cast(h, C.POINTER(C.c_uint8)), shape=(1 << 16,))
pin[:] = rng.integers(0, 256, 1 << 16, dtype=np.uint8)
d = C.c_void_p()
ok(lib.km_dev_alloc(ctx, 1 << 16, C.byref(d)), "dev_alloc")
ok(lib.km_upload_async(ctx, d, 256, h, 256, 256, 256), "upload_async (page-locked)")
ok(lib.km_upload_wait(ctx), "upload_wait")
back = np.zeros(1 << 16, np.uint8)
ok(lib.km_d2h(ctx, P(back), d, 1 << 16), "d2h")
assert np.array_equal(back, pin)
ok(lib.km_dev_free(ctx, d), "dev_free")
del pin
ok(lib.km_host_free(ctx, h), "host_free")

# ---- fine-grained mirrors: the stand-in Laplacian is the identity, so upload + download must reproduce the (strided) input
for shape in ((5, 7), (257, 1031), (1200, 900)):
    a = rng.integers(0, 256, shape, dtype=np.uint8)
    out = np.zeros_like(a)
    ok(lib.km_laplacian_u8(ctx, P(a), shape[0], shape[1], 7, P(out)), "laplacian")
    assert np.array_equal(out, a)
    eig = np.zeros(shape, np.float32)
    ok(lib.km_min_eigen(ctx, P(a), shape[0], shape[1], 15, P(eig)), "min_eigen")
    assert np.array_equal(eig, a.astype(np.float32))
err(lib.km_laplacian_u8(ctx, None, 5, 5, 7, P(out)), "laplacian null", KM_E_ARG)
err(lib.km_laplacian_u8(ctx, P(a), 0, 5, 7, P(out)), "laplacian empty", KM_E_ARG)
u16 = rng.integers(100, 9000, (333, 517)).astype(np.uint16)
v = u16[3:300, 10:500]
o8 = np.zeros(v.shape, np.uint8)
mm = (C.c_double * 2)()
ok(lib.km_to_uint8(ctx, P(v), 1, v.shape[0], v.shape[1], u16.strides[0] // 2, 0, P(o8), mm), "to_uint8 strided")
assert (mm[0], mm[1]) == (float(v.min()), float(v.max()))
err(lib.km_to_uint8(ctx, P(v), 9, v.shape[0], v.shape[1], u16.strides[0] // 2, 0, P(o8), mm), "to_uint8 dtype", KM_E_ARG)
err(lib.km_to_uint8(ctx, P(v), 1, v.shape[0], v.shape[1], 3, 0, P(o8), mm), "to_uint8 stride", KM_E_ARG)
sh = np.zeros_like(u16)
ok(lib.km_shift_image(ctx, P(u16), 2, u16.shape[0], u16.shape[1], u16.shape[1], 5, -7, P(sh)), "shift_image")
want = np.zeros_like(u16)
want[:-5, 7:] = u16[5:, :-7]
assert np.array_equal(sh, want)
err(lib.km_shift_image(ctx, P(u16), 3, 4, 4, 4, 0, 0, P(sh)), "shift elem size", KM_E_ARG)

# ---- ordering primitives on host buffers
keys = rng.integers(0, 1 << 62, 5000, dtype=np.uint64)
vals = np.arange(5000, dtype=np.uint32)
k2, v2 = keys.copy(), vals.copy()
ok(lib.km_sort_pairs_u64(ctx, P(k2), P(v2), 5000, 1), "sort_pairs")
o = np.argsort(keys, kind="stable")[::-1]
assert np.array_equal(k2, keys[o])
sc_in = rng.integers(0, 3, 7000).astype(np.uint32)
sc_out = np.zeros(7000, np.uint32)
ok(lib.km_exclusive_scan_u32(ctx, P(sc_in), P(sc_out), 7000, 0), "scan")
assert np.array_equal(sc_out, np.concatenate(([0], np.cumsum(sc_in)[:-1])).astype(np.uint32))

# ---- the blocking tile call on strided int16 boxes of changing size: workspace slots grow, every upload is checksummed
armed0 = C.c_int64()
for (H, W, cap) in ((64, 80, 50), (300, 420, 50), (120, 90, 80), (700, 900, 200)):
    mon = rng.integers(-4000, 8000, (H + 9, W + 30)).astype(np.int16)
    ref = rng.integers(-4000, 8000, (H + 9, W + 30)).astype(np.int16)
    mb, rb = mon[4:4 + H, 11:11 + W], ref[4:4 + H, 11:11 + W]
    mask = np.full((H, W), 255, np.uint8)
    p0, p1, p0r = (np.zeros((cap, 2), np.float32) for _ in range(3))
    n = C.c_int()
    prm = params(max_corners=cap)
    ok(lib.km_klt_tile(ctx, P(rb), P(mb), 2, H, W, mon.strides[0] // 2, mon.strides[0] // 2, P(mask), None, None, C.byref(prm), P(p0), P(p1), P(p0r), cap,
                       C.byref(n)), "klt_tile")
    assert 0 < n.value <= cap
    st = KltStats()
    ok(lib.km_get_klt_stats(ctx, C.byref(st)), "stats")
    assert (st.min_ref, st.max_ref, st.min_mon, st.max_mon) == (float(rb.min()), float(rb.max()), float(mb.min()), float(mb.max())), "the kernels saw the rasters as read"
    err(lib.km_klt_tile(ctx, P(rb), P(mb), 2, H, W, mon.strides[0] // 2, mon.strides[0] // 2, None, None, None, C.byref(prm), P(p0), P(p1), P(p0r), cap - 1,
                        C.byref(n)), "klt_tile capacity", KM_E_ARG)
    err(lib.km_klt_tile(ctx, P(rb), P(mb), 2, H, W, W - 1, W, None, None, None, C.byref(prm), P(p0), P(p1), P(p0r), cap, C.byref(n)), "klt_tile stride", KM_E_ARG)
bad = params(win_size=2)
err(lib.km_klt_tile(ctx, P(rb), P(mb), 2, H, W, W, W, None, None, None, C.byref(bad), P(p0), P(p1), P(p0r), cap, C.byref(n)), "klt_tile winSize", KM_E_ARG)
armed, missed = C.c_int64(), C.c_int64()
ok(lib.km_upload_check_stats(ctx, C.byref(armed), C.byref(missed)), "check_stats")
assert missed.value == 0 and (armed.value >= 12 if os.environ.get("KARIOS_HIP_UPLOAD_CHECKSUM") else armed.value == 0), (armed.value, missed.value)

# ---- allocation failure inside a call: an error code, no crash, the context stays usable
lib.stub_fail_malloc_after(0)
a = rng.integers(0, 256, (2000, 2100), dtype=np.uint8)      # larger than every slot so far: km_ws must allocate
out = np.zeros_like(a)
err(lib.km_laplacian_u8(ctx, P(a), 2000, 2100, 7, P(out)), "laplacian under allocation failure", KM_E_NOMEM)
assert b"hipMalloc" in lib.km_last_error(ctx)
ok(lib.km_laplacian_u8(ctx, P(a), 2000, 2100, 7, P(out)), "laplacian after the failure")
assert np.array_equal(out, a)

# ---- resident tiles: blocking frame call, the three-slot frame ring, the frame sink
H, W, cap = 400, 520, 64
mon = rng.integers(1, 9000, (H, W)).astype(np.uint16)
ref = rng.integers(1, 9000, (H, W)).astype(np.uint16)
dm, dr, sink = C.c_void_p(), C.c_void_p(), C.c_void_p()
ok(lib.km_dev_alloc(ctx, mon.nbytes, C.byref(dm)), "dev_alloc")
ok(lib.km_dev_alloc(ctx, ref.nbytes, C.byref(dr)), "dev_alloc")
ok(lib.km_h2d(ctx, dm, P(mon), mon.nbytes), "h2d")
ok(lib.km_h2d(ctx, dr, P(ref), ref.nbytes), "h2d")
prm = params(max_corners=cap)
blk_bytes = 16 + cap * 6 * 4 + cap * 8
block = np.zeros(blk_bytes // 4, np.float32)
ok(lib.km_klt_tile_frame_zncc_dev(ctx, dr, dm, 1, H, W, W, W, None, 0, None, None, C.byref(prm), 0.0, 0.0, dr, dm, H, W, W, W, 0.4, P(block), cap), "frame_zncc_dev")
hdr = block[:4].view(np.int32)
assert 0 < hdr[0] <= hdr[1] <= cap
err(lib.km_klt_tile_frame_zncc_dev(ctx, dr, dm, 1, H, W, W, W, None, 0, None, None, C.byref(prm), 0.0, 0.0, dr, dm, H, W, W, W, 0.4, None, cap), "frame null out", KM_E_ARG)
err(lib.km_klt_tile_frame_zncc_dev(ctx, dr, dm, 1, 1, 65536, 65536, 65536, None, 0, None, None, C.byref(prm), 0.0, 0.0, dr, dm, H, W, W, W, 0.4, P(block), cap),
    "frame of a 65536-column tile", KM_E_ARG)
ok(lib.km_dev_alloc(ctx, blk_bytes, C.byref(sink)), "dev_alloc sink")
ok(lib.km_set_frame_sink(ctx, sink, blk_bytes), "frame_sink")
tickets = []
for k in range(5):                                  # five submissions without a wait: slots are overwritten (the oldest is waited for inside)
    t = C.c_int(-1)
    ok(lib.km_klt_tile_frame_submit(ctx, dr, dm, 1, H, W, W, W, None, 0, None, None, C.byref(prm), float(k), 0.0, dr, dm, H, W, W, W, 0.4, cap, C.byref(t)), "submit")
    tickets.append(t.value)
assert tickets == [0, 1, 2, 0, 1]
blk, nb = C.c_void_p(), C.c_size_t()
for t in (2, 0, 1):
    assert lib.km_frame_wait(ctx, t, C.byref(blk), C.byref(nb)) == 0 and nb.value == blk_bytes
    got = np.ctypeslib.as_array(C.cast(blk, C.POINTER(C.c_float)), shape=(blk_bytes // 4,)).copy()
    assert got[:4].view(np.int32)[0] == hdr[0]
assert lib.km_frame_wait(ctx, 1, C.byref(blk), C.byref(nb)) < 0           # already collected
assert lib.km_frame_wait(ctx, 3, C.byref(blk), C.byref(nb)) < 0
sunk = np.zeros(blk_bytes // 4, np.float32)
ok(lib.km_d2h(ctx, P(sunk), sink, blk_bytes), "d2h sink")
assert np.array_equal(sunk[4:4 + cap], got[4:4 + cap])
ok(lib.km_set_frame_sink(ctx, sink, 8), "small sink")
t = C.c_int()
err(lib.km_klt_tile_frame_submit(ctx, dr, dm, 1, H, W, W, W, None, 0, None, None, C.byref(prm), 0.0, 0.0, dr, dm, H, W, W, W, 0.4, cap, C.byref(t)), "sink too small", KM_E_ARG)
ok(lib.km_set_frame_sink(ctx, None, 0), "sink off")
ok(lib.km_ctx_sync(ctx), "ctx_sync")

# ---- batched units (api_units.hip): three boxes of the resident pair in one submission, blocks into a pitched sink
es = 2
units = (KmUnit * 3)()
boxes = [(0, 0, 520, 300), (8, 40, 512, 200), (4, 100, 516, 300)]
for u, (bx, by, bw, bh) in zip(units, boxes):
    u.d_ref, u.d_mon, u.sref, u.smon, u.H, u.W = dr.value + (by * W + bx) * es, dm.value + (by * W + bx) * es, W, W, bh, bw
    u.d_ref_full, u.d_mon_full, u.sref_f, u.smon_f, u.Hf, u.Wf = dr.value, dm.value, W, W, H, W
    u.x_off, u.y_off = float(bx), float(by)
pitch = blk_bytes + 16
sink3 = C.c_void_p()
ok(lib.km_dev_alloc(ctx, 3 * pitch, C.byref(sink3)), "dev_alloc sink3")
ok(lib.km_set_frame_sink_pitch(ctx, sink3, 2 * pitch + blk_bytes, pitch), "sink pitch")
for k in range(5):
    t = C.c_int(-1)
    ok(lib.km_klt_units_frame_submit(ctx, units, 3, 1, None, None, C.byref(prm), 0.4, cap, C.byref(t)), "units submit")
    assert t.value == (k + 2) % 3, t.value                    # (the ring went on from the five single submissions above)
    assert lib.km_frame_wait(ctx, t.value, C.byref(blk), C.byref(nb)) == 0 and nb.value == 3 * blk_bytes
    got3 = np.ctypeslib.as_array(C.cast(blk, C.POINTER(C.c_float)), shape=(3 * blk_bytes // 4,)).copy().reshape(3, -1)
    h3 = got3[:, :4].view(np.int32)
    assert (h3[:, 1] == [40, 41, 42]).all() and (h3[:, 0] > 0).all() and (h3[:, 3] == [1000, 1001, 1002]).all(), h3
sunk3 = np.zeros(3 * pitch // 4, np.float32)
ok(lib.km_d2h(ctx, P(sunk3), sink3, 3 * pitch), "d2h sink3")
for k in range(3):
    assert np.array_equal(sunk3[k * pitch // 4:k * pitch // 4 + blk_bytes // 4].view(np.int32), got3[k].view(np.int32))
# ---- the same submissions as a SOFTWARE PIPELINE ("units_pipeline": two workspace lanes, the tail of a submission enqueued by the next
# one / km_frame_flush / any other entry point / km_ctx_sync), with and without a user mask per unit
mask_img = np.ones((H, W), np.uint8)
mask_img[50:90, 100:300] = 0
dmask = C.c_void_p()
ok(lib.km_dev_alloc(ctx, H * W, C.byref(dmask)), "dev_alloc mask")
ok(lib.km_h2d(ctx, dmask, P(mask_img), H * W), "h2d mask")
ok(lib.km_set_option(ctx, b"units_pipeline", 1), "pipeline on")
for masked in (False, True):
    for u, (bx, by, bw, bh) in zip(units, boxes):
        u.d_mask, u.smask = (dmask.value + by * W + bx, W) if masked else (None, 0)
    tickets = []
    for k in range(6):
        t = C.c_int(-1)
        ok(lib.km_klt_units_frame_submit(ctx, units, 3, 1, None, None, C.byref(prm), 0.4, cap, C.byref(t)), "units submit (pipelined)")
        tickets.append(t.value)
        if k == 3:
            ok(lib.km_frame_flush(ctx, t.value), "flush the newest")           # (its tail goes now; the next submission finds none deferred)
            ok(lib.km_frame_flush(ctx, t.value), "flush again: no-op")
        if k >= 1:                                                             # the previous submission: its tail travelled with this one
            assert lib.km_frame_wait(ctx, tickets[k - 1], C.byref(blk), C.byref(nb)) == 0 and nb.value == 3 * blk_bytes
            h3 = np.ctypeslib.as_array(C.cast(blk, C.POINTER(C.c_float)), shape=(3 * blk_bytes // 4,)).copy().reshape(3, -1)[:, :4].view(np.int32)
            assert (h3[:, 1] == [40, 41, 42]).all() and (h3[:, 0] > 0).all(), h3
    ok(lib.km_ctx_sync(ctx), "ctx_sync enqueues the last tail")
    assert lib.km_frame_wait(ctx, tickets[-1], C.byref(blk), C.byref(nb)) == 0 and nb.value == 3 * blk_bytes
t = C.c_int(-1)
ok(lib.km_klt_units_frame_submit(ctx, units, 3, 1, None, None, C.byref(prm), 0.4, cap, C.byref(t)), "units submit (tail deferred)")
ok(lib.km_klt_tile_frame_dev(ctx, dr, dm, 1, H, W, W, W, None, 0, None, None, C.byref(prm), 0.0, 0.0, P(np.zeros(4 + 6 * cap, np.float32)), cap), "another entry point enqueues the deferred tail first")
assert lib.km_frame_wait(ctx, t.value, C.byref(blk), C.byref(nb)) == 0
units[1].d_mask = None
err(lib.km_klt_units_frame_submit(ctx, units, 3, 1, None, None, C.byref(prm), 0.4, cap, C.byref(t)), "units: mixed masks", KM_E_ARG)
for u in units:
    u.d_mask, u.smask = None, 0
ok(lib.km_set_option(ctx, b"units_pipeline", 0), "pipeline off")
ok(lib.km_dev_free(ctx, dmask), "dev_free mask")
ok(lib.km_set_frame_sink_pitch(ctx, sink3, 2 * pitch, pitch), "sink too small for three")
err(lib.km_klt_units_frame_submit(ctx, units, 3, 1, None, None, C.byref(prm), 0.4, cap, C.byref(t)), "units: sink too small", KM_E_ARG)
ok(lib.km_set_frame_sink(ctx, None, 0), "sink off")
err(lib.km_klt_units_frame_submit(ctx, units, 0, 1, None, None, C.byref(prm), 0.4, cap, C.byref(t)), "units: none", KM_E_ARG)
err(lib.km_klt_units_frame_submit(ctx, units, 17, 1, None, None, C.byref(prm), 0.4, cap, C.byref(t)), "units: too many", KM_E_ARG)
err(lib.km_klt_units_frame_submit(ctx, None, 3, 1, None, None, C.byref(prm), 0.4, cap, C.byref(t)), "units: null", KM_E_ARG)
units[1].W = 300                                                # narrower than the 8-px eigenvalue kernel serves: not an error, "one by one"
assert lib.km_klt_units_frame_submit(ctx, units, 3, 1, None, None, C.byref(prm), 0.4, cap, C.byref(t)) == -4
units[1].W = 512
units[2].d_ref_full = None
err(lib.km_klt_units_frame_submit(ctx, units, 3, 1, None, None, C.byref(prm), 0.4, cap, C.byref(t)), "units: mixed score columns", KM_E_ARG)
ok(lib.km_ctx_sync(ctx), "ctx_sync")
ok(lib.km_dev_free(ctx, sink3), "dev_free sink3")

# ---- scores / filters on host key points
n = 300
x0 = rng.integers(0, W, n).astype(np.float32); y0 = rng.integers(0, H, n).astype(np.float32)
dx = rng.uniform(-1, 1, n).astype(np.float32); dy = rng.uniform(-1, 1, n).astype(np.float32)
z = np.zeros(n)
ok(lib.km_zncc_batch(ctx, P(ref), P(mon), 1, H, W, H, W, W, W, P(x0), P(y0), P(dx), P(dy), n, P(z)), "zncc_batch")
assert np.array_equal(z, (x0 + y0 + dx + dy).astype(np.float64))
s1, s2 = np.zeros(n), np.zeros(n)
ok(lib.km_mi_batch(ctx, P(ref), P(mon), 1, H, W, H, W, W, W, P(x0), P(y0), P(dx), P(dy), n, P(s1), P(s2)), "mi_batch")
assert np.array_equal(s1, x0.astype(np.float64)) and np.array_equal(s2, -x0.astype(np.float64))
err(lib.km_mi_batch(ctx, P(ref), P(mon), 1, H, W, H, W, W, W, P(x0), P(y0), P(dx), P(dy), n, None, None), "mi no output", KM_E_ARG)
keep = np.zeros(n, np.uint8)
nov = np.array([0.0, 65535.0])
ok(lib.km_dn_keep_dev(ctx, dr, dm, 1, H, W, W, W, P(x0), P(y0), n, P(nov), 2, None, None, P(keep)), "dn_keep")
assert keep.all()
x_bad = x0.copy(); x_bad[5] = W + 3
err(lib.km_dn_keep_dev(ctx, dr, dm, 1, H, W, W, W, P(x_bad), P(y0), n, P(nov), 2, None, None, P(keep)), "dn_keep outside", KM_E_ARG)
uv = rng.integers(30, 200, (n, 4)).astype(np.int32)
fl = np.zeros(n, np.uint8)
ok(lib.km_zncc_windows(ctx, P(ref), P(mon.astype(np.float64)), 1, 4, H, W, H, W, W, W, P(uv), 10, n, P(z), P(fl)), "zncc_windows")
q = rng.uniform(-0.02, 0.02, (64, 4)).astype(np.float32)
o = np.zeros(64, np.uint8)
ok(lib.km_lk_oscillation_probe(ctx, P(q), 64, P(o)), "oscillation probe")
# the host-side planning of the double-precision phase correlation (no device, no context): levels, Bluestein, position tables
for n_side, cols in ((10980, 0), (10980, 1), (1, 0), (7, 1), (3721, 0), (10007, 1), (65536, 1), (2 * 3 * 5 * 7 * 11 * 13, 0)):
    lv = (C.c_int * 32)()
    nl, bl = C.c_int(-1), C.c_int(-1)
    ng = np.full(n_side, -1, np.int32)
    assert lib.km_phase_plan(n_side, cols, lv, 16, C.byref(nl), C.byref(bl), P(ng)) == 0
    assert (bl.value > 0) == (n_side == 10007) and ng.min() >= 0 and ng.max() < n_side
assert lib.km_phase_plan(0, 0, lv, 16, None, None, None) < 0 and lib.km_phase_plan(10980, 0, lv, 1, None, None, None) < 0
rc_ = (C.c_double * 2)()
ok(lib.km_phase_shift(ctx, P(ref), P(mon), 1, H, W, W, W, rc_), "phase_shift")

for dptr in (dm, dr, sink):
    ok(lib.km_dev_free(ctx, dptr), "dev_free")
ok(lib.km_ctx_destroy(ctx), "ctx_destroy")
cnt = (C.c_long * 3)()
lib.stub_counters(cnt)
assert cnt[0] == 0 and cnt[1] == 0, f"HIP allocations left behind: device {cnt[0]}, page-locked {cnt[1]}"
assert cnt[2] == 0, f"{cnt[2]} asynchronous runtime copies touched PAGEABLE memory"
print("HOST-ASAN OK")
