"""The drop-in `karios_amd.matcher` classes on inputs the reference accepts but round 1 rejected or never tested:
`_zncc2` at any window size (the reference's own known-answer tests, tests/test_zncc_service.py:107-125), cross-sensor pixel
types, the kernel-size search combined with inverted / searched polarity on non-uint8 imagery."""
import itertools

import numpy as np
import pandas as pd
import pytest

from karios_amd import synth

pytestmark = pytest.mark.gpu


# ---------------------------------------------------------------------------- _zncc2, reference KATs
def test_zncc2_reference_known_answers(ops):
    from karios_amd.matcher.zncc_service import _zncc2
    ramp = np.arange(1, 10, dtype=np.float64).reshape(3, 3)
    assert abs(_zncc2(ramp, ramp.copy(), 1, 1, 1, 1, 1) - 1.0) < 1e-10            # test_zncc2_perfect_correlation
    assert _zncc2(ramp, 10 - ramp, 1, 1, 1, 1, 1) < 0                             # test_zncc2_anti_correlated
    assert abs(_zncc2(ramp, 10 - ramp, 1, 1, 1, 1, 1) + 1.0) < 1e-10
    flat = np.ones((5, 5))
    assert np.isnan(_zncc2(flat, flat, 2, 2, 2, 2, 2))                            # zero std -> NaN (test_zncc_zero_std_fix.py:65-70)
    assert np.isnan(_zncc2(ramp, ramp, 1, 1, 1, 1, 0))                            # one-pixel window: no variance
    with pytest.raises(IndexError):
        _zncc2(ramp, ramp, 0, 0, 0, 0, 1)
    with pytest.raises(ValueError, match="must be non-negative"):
        _zncc2(ramp, ramp, 1, 1, 1, 1, -1)


@pytest.mark.parametrize("half", [0, 1, 2, 7, 21, 30])
@pytest.mark.parametrize("dtypes", [(np.float64, np.float64), (np.uint8, np.uint16), (np.int16, np.float32), (np.int32, np.uint32),
                                    (np.uint16, np.uint16), (np.int64, np.float64)])
def test_zncc_windows_any_size_any_types_vs_numpy(ops, half, dtypes):
    """ops.zncc_windows == the numpy expression of `_zncc2` (zncc_service.py:111-126) on random windows."""
    rng = np.random.default_rng(half * 7 + 1)
    a = rng.integers(0, 200, (90, 120)).astype(dtypes[0])
    b = (rng.integers(0, 200, (101, 97)) + (rng.random((101, 97)) if np.dtype(dtypes[1]).kind == "f" else 0)).astype(dtypes[1])
    a[40:80, 50:110] = 7                                  # a flat region: windows inside have no variance
    n = 60
    u1, v1 = rng.integers(-2, 92, n), rng.integers(-2, 122, n)
    u2, v2 = rng.integers(-2, 103, n), rng.integers(-2, 99, n)
    got, outside = ops.zncc_windows(a, b, u1, v1, u2, v2, half)
    for k in range(n):
        inside = half <= u1[k] < 90 - half and half <= v1[k] < 120 - half and half <= u2[k] < 101 - half and half <= v2[k] < 97 - half
        assert outside[k] == (not inside)
        if not inside:
            assert np.isnan(got[k])
            continue
        p1 = a[u1[k] - half:u1[k] + half + 1, v1[k] - half:v1[k] + half + 1].astype(np.float64)
        p2 = b[u2[k] - half:u2[k] + half + 1, v2[k] - half:v2[k] + half + 1].astype(np.float64)
        s1, s2 = p1.std(), p2.std()
        if s1 == 0 or s2 == 0:
            assert np.isnan(got[k])
        else:
            want = np.mean(((p1 - p1.mean()) / s1) * ((p2 - p2.mean()) / s2))
            assert abs(got[k] - want) <= 1e-9


def test_zncc_service_cross_sensor_types_and_float64_frames(ops, O):
    """uint8 reference against uint16 monitored data, and a frame that carries a float64 column (then the reference's
    `x0 + dx` is a float64 sum): same values as the single-type kernel on lossless casts / as the oracle."""
    from karios_amd.core import NumpyRasterImage
    from karios_amd.matcher import MutualInfoService, ZNCCService
    mon, ref = synth.make_pair(300, 320, 0.6, -0.4, seed=3)
    ref8 = (ref >> 6).astype(np.uint8)
    rng = np.random.default_rng(5)
    n = 200
    df = pd.DataFrame({"x0": rng.integers(0, 320, n).astype(np.float32), "y0": rng.integers(0, 300, n).astype(np.float32),
                       "dx": rng.uniform(-2, 2, n).astype(np.float32), "dy": rng.uniform(-2, 2, n).astype(np.float32)})
    df.loc[:20, "dx"] = np.float32(0.5)                   # x.5 sums: half-to-even
    want = O.zncc_batch(ref8.astype(np.uint16), mon, *(df[c].to_numpy() for c in ("x0", "y0", "dx", "dy")))
    got = ZNCCService().compute_zncc(df, NumpyRasterImage(mon), NumpyRasterImage(ref8)).to_numpy()
    assert np.array_equal(np.isnan(got), np.isnan(want)) and np.nanmax(np.abs(got - want)) <= 1e-9
    wide = df.assign(extra=np.arange(n, dtype=np.float64))
    got64 = ZNCCService().compute_zncc(wide, NumpyRasterImage(mon), NumpyRasterImage(ref)).to_numpy()
    want64 = O.zncc_batch(ref, mon, *(df[c].to_numpy() for c in ("x0", "y0", "dx", "dy")))
    same = np.isclose(np.rint(df.x0.astype(np.float64) + df.dx.astype(np.float64)), np.rint(df.x0 + df.dx)) & \
        np.isclose(np.rint(df.y0.astype(np.float64) + df.dy.astype(np.float64)), np.rint(df.y0 + df.dy))
    assert np.array_equal(np.isnan(got64[same]), np.isnan(want64[same])) and np.nanmax(np.abs(got64[same] - want64[same])) <= 1e-9
    st = MutualInfoService().compute_mutual_info(df, NumpyRasterImage(mon), NumpyRasterImage(ref8)).to_numpy()
    est, _ = O.mi_batch(ref8.astype(np.uint16), mon, *(df[c].to_numpy() for c in ("x0", "y0", "dx", "dy")))
    assert np.array_equal(np.isnan(st), np.isnan(est)) and np.nanmax(np.abs(st - est)) <= 1e-9


# ---------------------------------------------------------------------------- KLT.match on pixel types round 1 rejected
def _oracle_match(O, mon, ref, conf, nodata=(None, None), invert=False):
    """Reference semantics for any pixel types: mask from the raw values, each image stretched on its own (klt.py:42-49,
    268-273), then the uint8 pipeline."""
    mask = (mon != 0) & (ref != 0) & np.isfinite(ref) & np.isfinite(mon)
    if nodata[0] is not None:
        mask &= mon != nodata[0]
    if nodata[1] is not None:
        mask &= ref != nodata[1]

    def stretch(a):
        if a.dtype == np.uint8:
            return a
        lo, hi = float(np.nanmin(a)), float(np.nanmax(a))
        return ((a - lo) / (hi - lo) * 255).astype(np.uint8) if hi > lo else np.zeros(a.shape, np.uint8)

    return O.klt_tile(stretch(mon), stretch(ref), conf, mask_box=mask.astype(np.uint8), invert_mon=invert)


@pytest.mark.parametrize("types", [(np.uint16, np.uint8), (np.uint8, np.uint16), (np.int32, np.int32), (np.float64, np.uint16),
                                   (np.uint32, np.float32)])
def test_klt_match_accepts_cross_sensor_and_wide_pixel_types(ops, O, types):
    from karios_amd.core import KLTConfiguration, NumpyRasterImage
    from karios_amd.matcher import KLT
    mon, ref = synth.make_pair(260, 300, 0.4, 0.3, seed=21, nodata_wedge=True)

    def typed(a, t):
        if t == np.uint8:
            return (a >> 6).astype(np.uint8)
        if t == np.int32:
            return (a.astype(np.int64) * 1000 - 5_000_000 * (a > 0)).astype(np.int32)     # negative values too, 0 stays no-data
        if t == np.uint32:
            return (a.astype(np.int64) * 70000).astype(np.uint32)
        if t == np.float64:
            return np.where(a > 0, a * 3.5 + 0.25, 0.0)
        return a.astype(t)

    mon_t, ref_t = typed(mon, types[0]), typed(ref, types[1])
    conf = KLTConfiguration(maxCorners=400, laplacian_kernel_size=5)
    frames = list(KLT(conf).match(NumpyRasterImage(mon_t), NumpyRasterImage(ref_t), None))
    exp = _oracle_match(O, mon_t, ref_t, conf)
    assert len(frames) == 1 and len(frames[0]) == len(exp["x0"]) > 50
    for col in ("x0", "y0", "dx", "dy", "score"):
        np.testing.assert_array_equal(frames[0][col].to_numpy(), exp[col])


@pytest.mark.parametrize("polarity", [True, "auto"])
@pytest.mark.parametrize("outliers", [False, True])
def test_kernel_search_with_inverted_and_searched_polarity_on_uint16(ops, O, polarity, outliers):
    """laplacian_kernel_size='auto' x laplacian_invert_polarity in {True, 'auto'} on uint16 imagery (klt.py:419-429, 475-476):
    the winner, its polarity and its frame equal the reference's procedure carried out with the oracle."""
    from karios_amd.core import KLTConfiguration, NumpyRasterImage
    from karios_amd.matcher import KLT
    from oracle import oracle as Om
    mon, ref = synth.make_cross_sensor_pair(200, 240, seed=4)[:2]
    conf = KLTConfiguration(maxCorners=300, minDistance=6, blocksize=7, laplacian_kernel_size="auto", laplacian_invert_polarity=polarity,
                            outliers_filtering=outliers)
    cands = [3, 5, 7, 9, 11]
    mask, _ = O.auto_mask(mon, ref)

    def search(invert):
        laps_m = {k: O.laplacian_u8(O.to_uint8(mon, invert=invert), k) for k in cands}
        laps_r = {k: O.laplacian_u8(O.to_uint8(ref), k) for k in cands}
        p0s = {k: O.good_features(laps_r[k], mask, conf.maxCorners, conf.qualityLevel, conf.minDistance, conf.blocksize) for k in cands}
        best, best_ratio, best_res = None, -1.0, None
        for mk, rk in itertools.product(cands, repeat=2):
            res = None if p0s[rk] is None else O.klt_tracker(laps_r[rk], laps_m[mk], mask, conf, p0=p0s[rk])
            if res is None:
                continue
            ratio = len(res[0]["x0"]) / res[1] if res[1] else 0.0
            if ratio > best_ratio:
                best_ratio, best, best_res = ratio, (mk, rk), res
        return best, best_ratio, best_res

    runs = {"inverted": search(True)} if polarity is True else {"normal": search(False), "inverted": search(True)}
    label = max(runs, key=lambda k: runs[k][1])          # first maximum: 'normal' keeps a tie
    best, _, (pts, _ninit) = runs[label]
    klt = KLT(conf)
    frames = list(klt.match(NumpyRasterImage(mon), NumpyRasterImage(ref), None))
    assert klt.auto_selected_ksize == best
    assert klt.auto_selected_polarity == (label if polarity == "auto" else None)
    order = np.lexsort((pts["y0"], pts["x0"]))
    assert len(frames) == 1 and len(frames[0]) == len(order)
    for col in ("x0", "y0", "dx", "dy", "score"):
        np.testing.assert_array_equal(frames[0][col].to_numpy(), np.asarray(pts[col], np.float32)[order])


# ---------------------------------------------------------------------------- hand-written float32 phase correlation
@pytest.fixture(params=[1, 0], ids=["rows_61xM_wave_local", "rows_stockham"])
def fft_form(request, ops):
    """Both row kernels of k_fft.hip for lengths 61 * M: the wave-local form (default) and the workgroup-wide Stockham kernel
    (every other length, and 61 * M with M > 192)."""
    ctx = ops._lib.default_context()
    ctx.set_option("fft61", request.param)
    yield request.param
    ctx.set_option("fft61", 1)


@pytest.mark.parametrize("shape", [(244, 183), (420, 360), (1098, 1220), (96, 250), (366, 366), (61, 122), (1830, 700), (122, 3721),
                                   (427, 915), (2135, 128), (64, 5490), (10980, 61), (3660, 854)])
def test_fast_phase_correlation_equals_double_precision_path(ops, O, shape, fft_form):
    """k_fft.hip (float32, radices 2/3/4/5/7/61; rows of length 61 * M as 61-point transforms + M-point transforms per wavefront)
    against the double-precision path (k_fft64.hip, hand-written as well) and the oracle: same integer shifts on shifted copies (clear peak -> fast path is
    trusted), and the double path takes over when the peak is split evenly."""
    from karios_amd._lib import default_context
    H, W = shape
    ctx = default_context()
    base, _ = synth.make_pair(H + 80, W + 80, 0.0, 0.0, seed=H + W, noise_sigma=0.0)
    rng = np.random.default_rng(H * 7 + W)
    for trial in range(4):
        sy, sx = int(rng.integers(-min(30, H // 8), min(30, H // 8) + 1)), int(rng.integers(-min(30, W // 8), min(30, W // 8) + 1))
        a = base[40:40 + H, 40:40 + W]
        b = base[40 - sy:40 - sy + H, 40 - sx:40 - sx + W]
        if trial == 3:
            a, b = (a >> 6).astype(np.uint8), (b >> 6).astype(np.uint8)
        got = ops.phase_cross_correlation(b, a)
        path, margin = ctx.phase_info()
        assert path == 1 and margin > 0.05, (path, margin)
        ctx.set_option("phase_fp64", 1)
        try:
            ref64 = ops.phase_cross_correlation(b, a)
            assert ctx.phase_info()[0] == 2
        finally:
            ctx.set_option("phase_fp64", 0)
        np.testing.assert_array_equal(got, ref64)
        np.testing.assert_array_equal(got, O.phase_cross_correlation(b, a))
        if min(H, W) >= 200:                      # (a small image shifted by a good part of its size has no reliable peak)
            np.testing.assert_array_equal(got, [sy, sx])


@pytest.mark.parametrize("shape", [(244, 183), (366, 366), (1098, 1220), (61, 122), (427, 915), (183, 1220), (1830, 854)])
def test_hermitian_inverse_of_the_float32_phase_correlation_equals_the_full_plane_form(ops, O, shape):
    """`fft_herm` (default): the inverse transform runs on the half plane kx <= W / 2 and packs two image rows into one complex
    transform (k_fft.hip, mode 5) - even and odd sides, an odd number of rows (the last pair has one member): same integer shifts as
    the full-plane form and the oracle (large_offset.py:39), peak margins within 1e-3 of each other."""
    from karios_amd._lib import default_context
    H, W = shape
    ctx = default_context()
    base, _ = synth.make_pair(H + 80, W + 80, 0.0, 0.0, seed=H + 3 * W, noise_sigma=0.0)
    rng = np.random.default_rng(H + W)
    try:
        for trial in range(3):
            sy, sx = int(rng.integers(-min(30, H // 8), min(30, H // 8) + 1)), int(rng.integers(-min(30, W // 8), min(30, W // 8) + 1))
            a = base[40:40 + H, 40:40 + W]
            b = base[40 - sy:40 - sy + H, 40 - sx:40 - sx + W]
            res = {}
            for herm in (1, 0):
                ctx.set_option("fft_herm", herm)
                res[herm] = (ops.phase_cross_correlation(b, a), ctx.phase_info())
                assert res[herm][1][0] == 1
            np.testing.assert_array_equal(res[1][0], res[0][0])
            np.testing.assert_array_equal(res[1][0], O.phase_cross_correlation(b, a))
            assert abs(res[1][1][1] - res[0][1][1]) < 1e-3
    finally:
        ctx.set_option("fft_herm", 1)


@pytest.mark.parametrize("shape", [(61, 45), (96, 130), (1, 300), (300, 1), (2, 2), (3, 5), (11, 13), (127, 254), (131, 200), (200, 131), (257, 263),
                                   (1000, 1009), (4096, 64), (64, 4096), (4099, 37), (37, 4099), (512, 6000), (2135, 128), (122, 3721), (1830, 1098)])
def test_double_precision_phase_correlation_for_every_kind_of_side(ops, O, shape):
    """k_fft64.hip (complex128 like the reference, large_offset.py:39; hand-written - no FFT library is linked): sides with only small
    prime factors (levels in LDS), with a prime of 11 .. 127 such as Sentinel-2's 61 (a level kernel of its own), with larger primes
    (Bluestein's chirp-z along that dimension), one-pixel-wide images, and sides that need three levels - same integer shifts as the
    oracle, on every pixel type, with the image loads / the arg-max fused into the first / last level and as passes of their own."""
    from karios_amd._lib import default_context
    H, W = shape
    ctx = default_context()
    base, _ = synth.make_pair(H + 80, W + 80, 0.0, 0.0, seed=H * 3 + W, noise_sigma=0.0)
    rng = np.random.default_rng(H * 7 + W)
    ctx.set_option("phase_fp64", 1)
    try:
        for trial, dtype in enumerate((np.uint16, np.uint8, np.int16, np.float32)):
            sy = int(rng.integers(-min(30, H // 8), min(30, H // 8) + 1))
            sx = int(rng.integers(-min(30, W // 8), min(30, W // 8) + 1))
            a = base[40:40 + H, 40:40 + W]
            b = base[40 - sy:40 - sy + H, 40 - sx:40 - sx + W]
            if dtype == np.uint8:
                a, b = (a >> 6).astype(np.uint8), (b >> 6).astype(np.uint8)
            elif dtype == np.int16:
                a, b = (a.astype(np.int32) - 9000).astype(np.int16), (b.astype(np.int32) - 9000).astype(np.int16)
            elif dtype == np.float32:
                a, b = a.astype(np.float32) * np.float32(0.25), b.astype(np.float32) * np.float32(0.25)
            want = O.phase_cross_correlation(b, a)
            # pair: two image rows per transform in the inverse along the rows; half: only the columns kx <= W / 2 through the inverse column levels
            for plain, pair, half in ((0, 1, 1), (0, 1, 0), (0, 0, 0), (1, 1, 1)):
                ctx.set_option("f64_plain", plain)
                ctx.set_option("f64_pair", pair)
                ctx.set_option("f64_half", half)
                got = ops.phase_cross_correlation(b, a)
                assert ctx.phase_info()[0] == 2
                np.testing.assert_array_equal(got, want, err_msg=f"{shape} {dtype.__name__} plain={plain} pair={pair} half={half}")
            if min(H, W) >= 200:
                np.testing.assert_array_equal(got, [sy, sx])
    finally:
        ctx.set_option("phase_fp64", 0)
        ctx.set_option("f64_plain", 0)
        ctx.set_option("f64_pair", 1)
        ctx.set_option("f64_half", 1)


def test_phase_correlation_survives_toggling_the_row_form_on_one_context(ops, O):
    """the twiddle cache of k_fft.hip was keyed on the row length only, but the table's layout depends on the
    plan (the 61 * M form appends two sub-tables) - the SAME shape with fft61 = 0 and then fft61 = 1 on one context returned early
    with never-uploaded tables.  Same shape, both orders, several sides that both forms accept."""
    from karios_amd._lib import Context
    for (H, W) in ((366, 366), (1098, 1220), (427, 915)):
        base, _ = synth.make_pair(H + 80, W + 80, 0.0, 0.0, seed=H + W + 1, noise_sigma=0.0)
        a = base[40:40 + H, 40:40 + W]
        b = base[40 - 9:40 - 9 + H, 40 + 13:40 + 13 + W]
        want = O.phase_cross_correlation(b, a)
        for order in ((0, 1, 0, 1), (1, 0, 1)):
            ctx = Context()
            try:
                for form in order:
                    ctx.set_option("fft61", form)
                    got = ops.phase_cross_correlation(b, a, ctx=ctx)
                    assert ctx.phase_info()[0] == 1            # the float32 path answered (no double-precision rescue)
                    np.testing.assert_array_equal(got, want)
            finally:
                ctx.close()


@pytest.mark.parametrize("dtype", [np.uint16, np.int16, np.uint8, np.float32])
def test_frames_scored_with_both_mutual_information_columns_in_the_tile_call(ops, O, dtype):
    """`KariosAPI._handle_klt_results` (api/core.py:894-907) scores every confident candidate with ZNCC, `mutual_info_score`
    (mutual_info_service.py:73-130) and `mi_score` (zncc_service.py:240-287); with `mutual_info=True` all three ride in the tile's
    device call.  Blocking call, submitted call and FrameStream give the same block; the columns equal the oracle's on the frame's
    key points (<= 1e-9, NaN exactly where score < threshold or the chip leaves the image); the ZNCC-only frame is unchanged."""
    from karios_amd.core import KLTConfiguration
    from karios_amd.resident import ResidentPair
    from karios_amd.stream import FrameStream
    from karios_amd import results
    mon, ref = synth.make_pair(640, 760, 0.4, -0.3, seed=31, nodata_wedge=True)
    if dtype == np.uint8:
        mon, ref = (mon >> 5).astype(np.uint8), (ref >> 5).astype(np.uint8)
    elif dtype == np.int16:
        mon, ref = (mon.astype(np.int32) - 4000).astype(np.int16), (ref.astype(np.int32) - 4000).astype(np.int16)
    elif dtype == np.float32:
        mon, ref = mon.astype(np.float32) * np.float32(0.37), ref.astype(np.float32) * np.float32(0.37)
    pair = ResidentPair.upload(mon, ref)
    conf = KLTConfiguration(maxCorners=1500)
    thr = float(np.float32(np.median(pair.match_tile(conf)["score"].to_numpy())))       # half of the rows are scored, half are not
    f1 = pair.match_tile(conf, zncc_threshold=thr)
    f3 = pair.match_tile(conf, zncc_threshold=thr, mutual_info=True)
    assert list(f3.columns) == ["x0", "y0", "dx", "dy", "score", "zncc_score", "mutual_info_score", "mi_score"]
    pd.testing.assert_frame_equal(f3[list(f1.columns)], f1, check_exact=True)
    x0, y0, dx, dy, sc = (f3[c].to_numpy() for c in ("x0", "y0", "dx", "dy", "score"))
    keep = sc >= np.float32(thr)
    assert keep.sum() > 200 and (~keep).sum() > 200
    st, nmi = O.mi_batch(ref, mon, x0[keep], y0[keep], dx[keep], dy[keep])
    for col, want in (("mutual_info_score", st), ("mi_score", nmi)):
        got = f3[col].to_numpy()
        assert np.all(np.isnan(got[~keep]))
        assert np.array_equal(np.isnan(got[keep]), np.isnan(want))
        assert np.nanmax(np.abs(got[keep] - want)) <= 1e-9
    assert np.isnan(st).any() or True                      # (key points near the wedge / border: NaN by the bounds rule)
    # submitted form == blocking form, bit for bit
    raw = pair.submit_tile(conf, zncc_threshold=thr, mutual_info=True).result()
    assert raw.with_zncc == 3
    pd.testing.assert_frame_equal(raw.to_frame(), f3, check_exact=True)
    # FrameStream: the whole of _handle_klt_results' scoring columns, no second pass
    with FrameStream(thr, depth=1, mutual_info=True) as s:
        got = s.submit(pair, conf) + s.drain()
    fs = got[0].frame
    assert {"radial error", "angle", "zncc_score", "mutual_info_score", "mi_score"} <= set(fs.columns)
    for col in ("zncc_score", "mutual_info_score", "mi_score"):
        np.testing.assert_array_equal(fs[col].to_numpy().view(np.int64), f3[col].to_numpy().view(np.int64))
    # ... and the host-side scorer leaves device-scored columns alone (results.handle_klt_results reorders, never recomputes)
    again = pair.score_frame(f3.copy(), thr, mutual_info=True)
    np.testing.assert_array_equal(again["mi_score"].to_numpy().view(np.int64), f3["mi_score"].to_numpy().view(np.int64))
    assert list(again[results.CSV_COLUMNS].columns) == results.CSV_COLUMNS
