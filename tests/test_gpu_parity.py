"""GPU parity: every HIP operator against the CPU oracle on the same seeded inputs,
called through the C ABI (karios_amd.ops -> libkarios_hip.so).

Bars (BASELINE.json north_star / SURVEY.md 8d): integer / byte / index results bit-exact;
sub-pixel displacements within 1e-3 px; ZNCC within 1e-9.
"""
import numpy as np
import pytest

from conftest import rand_u8
from karios_amd import synth

pytestmark = pytest.mark.gpu

SHAPES = [(64, 64), (67, 93), (200, 131), (33, 257), (5, 7), (1, 40), (40, 1), (130, 64)]


@pytest.mark.parametrize("dtype", [np.uint16, np.int16, np.float32, np.uint8])
@pytest.mark.parametrize("invert", [False, True])
def test_to_uint8_bit_exact(ops, O, dtype, invert):
    rng = np.random.default_rng(1)
    for shape in [(67, 93), (128, 256), (5, 3)]:
        if dtype == np.float32:
            a = (rng.standard_normal(shape) * 1000).astype(np.float32)
            a[0, 0] = np.nan
        elif dtype == np.int16:
            a = rng.integers(-3000, 9000, shape).astype(np.int16)
        elif dtype == np.uint8:
            a = rng.integers(0, 256, shape).astype(np.uint8)
        else:
            a = rng.integers(1, 16000, shape).astype(np.uint16)
        got, mm = ops.to_uint8(a, invert=invert, return_minmax=True)
        exp = O.to_uint8(a, invert=invert)
        np.testing.assert_array_equal(got, exp)
        if dtype != np.uint8:
            assert mm == O.minmax(a)


def test_to_uint8_constant_and_strided(ops, O):
    a = np.full((40, 50), 1234, np.uint16)
    np.testing.assert_array_equal(ops.to_uint8(a), np.zeros_like(a, np.uint8))
    big = np.random.default_rng(2).integers(0, 5000, (100, 300)).astype(np.uint16)
    view = big[10:90, 17:203]  # row stride > width, unaligned start
    np.testing.assert_array_equal(ops.to_uint8(view), O.to_uint8(np.ascontiguousarray(view)))


@pytest.mark.parametrize("dtype", [np.uint16, np.float32])
def test_auto_mask(ops, O, dtype):
    rng = np.random.default_rng(3)
    mon = rng.integers(0, 4, (70, 90)).astype(dtype)
    ref = rng.integers(0, 4, (70, 90)).astype(dtype)
    if dtype == np.float32:
        mon[3, 3] = np.nan
        ref[4, 4] = np.inf
    for nd in [(None, None), (2, None), (None, 3), (1, 2)]:
        got, nv = ops.auto_mask(mon, ref, *nd)
        exp, ev = O.auto_mask(mon, ref, *nd)
        np.testing.assert_array_equal(got, exp)
        assert nv == ev


def _prefilter_case(dtype, shape, rng_span, seed):
    """Image pair holding every value of [mn, mn + span] (all exact multiples of the stretch included) plus zeros."""
    rng = np.random.default_rng(seed)
    H, W = shape
    n = H * W
    mn = 0 if np.dtype(dtype).kind == "u" else -min(rng_span // 2, 30000)
    if np.dtype(dtype) == np.float32:
        ref = (rng.random((H, W)) * rng_span).astype(np.float32)
        mon = (rng.random((H, W)) * rng_span * 0.7 + 3).astype(np.float32)
        ref[rng.random((H, W)) < 0.01] = np.nan
    else:
        ref = ((np.arange(n) % (rng_span + 1)) + mn).astype(dtype).reshape(H, W)
        mon = ((rng.permutation(n) % (rng_span + 1)) + mn).astype(dtype).reshape(H, W)
    ref[rng.random((H, W)) < 0.02] = 0
    mon[rng.random((H, W)) < 0.02] = 0
    return ref, mon


@pytest.mark.parametrize("dtype,span", [(np.uint16, 65535), (np.uint16, 510), (np.uint16, 12345), (np.uint16, 200), (np.int16, 65535),
                                        (np.int16, 1020), (np.uint8, 255), (np.float32, 4000)])
# widths that are no multiple of 4: the marching kernel's last strip is shifted to end at the right edge and counts valid pixels only
# beyond its left neighbour's columns (517, 1243); 746 = 3 x 248 + 2 leaves it fewer than 4 columns, 253 is below the shift's minimum
@pytest.mark.parametrize("shape", [(300, 520), (203, 517), (70, 1000), (64, 746), (90, 1243), (50, 253)])
def test_tile_prefilter_bit_exact(ops, O, dtype, span, shape):
    """Fused stretch + Laplacian (both images) + auto mask == the three reference steps done one by one (klt.py:268-273, 407-436)."""
    if np.dtype(dtype) == np.uint8:
        span = 255
    ref, mon = _prefilter_case(dtype, shape, span, seed=span + shape[1])
    cases = [dict(kr=7, km=7, inv=False, nr=None, nm=None), dict(kr=3, km=5, inv=True, nr=None, nm=None),
             dict(kr=1, km=7, inv=False, nr=float(ref[5, 7]), nm=float(mon[9, 3])), dict(kr=7, km=3, inv=True, nr=0.5, nm=1e9),
             dict(kr=9, km=11, inv=False, nr=-7.0, nm=None)]
    for cs in cases:
        if np.dtype(dtype) == np.float32 and cs["nr"] is not None and np.isnan(cs["nr"]):
            continue
        lr, lm, mk, nv = ops.tile_prefilter(ref, mon, nodata_ref=cs["nr"], nodata_mon=cs["nm"], ref_ksize=cs["kr"], mon_ksize=cs["km"],
                                            invert_mon=cs["inv"])
        want_r = O.laplacian_u8(O.to_uint8(ref), cs["kr"])
        want_m = O.laplacian_u8(O.to_uint8(mon, invert=cs["inv"]), cs["km"])
        want_mask, want_valid = O.auto_mask(mon, ref, nodata_mon=cs["nm"], nodata_ref=cs["nr"])
        assert np.array_equal(lr, want_r), cs
        assert np.array_equal(lm, want_m), cs
        assert np.array_equal(mk, want_mask), cs
        assert nv == want_valid, cs
        lr2, lm2, mk2, nv2 = ops.tile_prefilter(ref, mon, ref_ksize=cs["kr"], mon_ksize=cs["km"], invert_mon=cs["inv"], with_mask=False)
        assert mk2 is None and nv2 is None and np.array_equal(lr2, want_r) and np.array_equal(lm2, want_m)


@pytest.mark.parametrize("ksize", [1, 3, 5, 7, 9, 11])
def test_laplacian_bit_exact(ops, O, ksize):
    for i, shape in enumerate(SHAPES):
        img = rand_u8(shape, seed=10 + i)
        np.testing.assert_array_equal(ops.laplacian_u8(img, ksize), O.laplacian_u8(img, ksize), err_msg=str(shape))
    smooth = O.to_uint8(synth.make_pair(150, 170)[1])
    np.testing.assert_array_equal(ops.laplacian_u8(smooth, ksize), O.laplacian_u8(smooth, ksize))


def test_laplacian_rejects_bad_ksize(ops):
    with pytest.raises(ops.KariosHipError):
        ops.laplacian_u8(rand_u8((20, 20)), 4)
    with pytest.raises(ops.KariosHipError):
        ops.laplacian_u8(rand_u8((20, 20)), 13)


@pytest.mark.parametrize("block", [3, 7, 15, 4, 1])
def test_min_eigen_bit_exact(ops, O, block):
    for i, shape in enumerate(SHAPES):
        img = rand_u8(shape, seed=30 + i)
        got, exp = ops.min_eigen(img, block), O.min_eigen(img, block)
        np.testing.assert_array_equal(got.view(np.uint32), exp.view(np.uint32), err_msg=f"{shape} block={block}")
    lap = O.laplacian_u8(O.to_uint8(synth.make_pair(200, 260)[1]), 7)
    np.testing.assert_array_equal(ops.min_eigen(lap, block).view(np.uint32), O.min_eigen(lap, block).view(np.uint32))


def test_pyrdown_bit_exact(ops, O):
    # (widths around the work split of the kernel: interior quads 1 .. (W - 12) / 8 in clean waves, the border quads gathered; W < 28 has
    #  no interior; 2073 = more than 256 interior quads; heights that are no multiple of the 8-row item)
    for i, shape in enumerate(SHAPES + [(255, 300), (9, 27), (17, 28), (33, 29), (40, 35), (21, 36), (70, 2073), (19, 2061), (3, 5), (2, 2), (131, 4099)]):
        img = rand_u8(shape, seed=50 + i)
        np.testing.assert_array_equal(ops.pyr_down(img), O.pyrdown_u8(img), err_msg=str(shape))


def _lap_pair(O, H, W, sx=0.5, sy=0.25, k=7, seed=20260101, **kw):
    mon, ref = synth.make_pair(H, W, sx, sy, seed=seed, **kw)
    return O.laplacian_u8(O.to_uint8(mon), k), O.laplacian_u8(O.to_uint8(ref), k), mon, ref


@pytest.mark.parametrize("params", [
    dict(maxCorners=2000, qualityLevel=0.1, minDistance=10, blockSize=15),
    dict(maxCorners=0, qualityLevel=0.05, minDistance=5, blockSize=7),
    dict(maxCorners=50, qualityLevel=0.01, minDistance=3.5, blockSize=3),
    dict(maxCorners=300, qualityLevel=0.2, minDistance=0, blockSize=5),
    dict(maxCorners=100000, qualityLevel=0.001, minDistance=1, blockSize=3),
])
def test_good_features_indices_bit_exact(ops, O, params):
    lap_mon, lap_ref, mon, ref = _lap_pair(O, 300, 420)
    mask, _ = O.auto_mask(mon, ref)
    for m in (None, mask):
        got = ops.good_features_to_track(lap_ref, mask=m, **params)
        exp = O.good_features(lap_ref, mask=m, **params)
        assert (got is None) == (exp is None)
        if exp is not None:
            np.testing.assert_array_equal(got, exp)  # same corners, same (strength) order


def test_good_features_masked_wedge_and_ties(ops, O):
    lap_mon, lap_ref, mon, ref = _lap_pair(O, 256, 256, nodata_wedge=True)
    mask, _ = O.auto_mask(mon, ref)
    got = ops.good_features_to_track(lap_ref, 5000, 0.1, 10, mask=mask, blockSize=15)
    exp = O.good_features(lap_ref, mask, 5000, 0.1, 10, 15)
    np.testing.assert_array_equal(got, exp)
    # periodic pattern: exact ties everywhere, order decided by the raster-index rule
    tile = np.zeros((16, 16), np.uint8)
    tile[4:9, 4:9] = 255
    img = np.tile(tile, (12, 14))
    got = ops.good_features_to_track(img, 0, 0.01, 8, blockSize=5)
    exp = O.good_features(img, None, 0, 0.01, 8, 5)
    np.testing.assert_array_equal(got, exp)


@pytest.mark.parametrize("block", [3, 15])
def test_good_features_plateau_falls_back_from_fused_kernel(ops, O, block):
    """img = g(x) + h(y) with periods dividing blockSize: the box sums of dx^2 and dy^2 cover whole periods (constant), the
    one of dx*dy vanishes, so lambda is one constant over the whole interior and EVERY pixel ties as a 3x3 maximum.  A row
    group then overflows the candidate stage of the fused eig+candidate kernel; the library must notice and repeat the
    pass with the eig-map + candidate kernels (same result as the oracle: ties resolved by the raster-index rule)."""
    from karios_amd._lib import default_context
    g = np.array([0, 60, 200, 90, 30] if block == 15 else [0, 90, 200])
    h = np.array([0, 50, 20])
    img = (g[np.arange(700) % len(g)][None, :] + h[np.arange(330) % 3][:, None]).astype(np.uint8)
    exp = O.good_features(img, None, 2000, 0.01, 4, block)
    ctx = default_context()
    from karios_amd._lib import PATH_STAGE_FALLBACK
    for fused in (1, 0):
        ctx.set_option("fused_eig", fused)
        try:
            got = ops.good_features_to_track(img, 2000, 0.01, 4, blockSize=block)
        finally:
            ctx.set_option("fused_eig", 1)
        np.testing.assert_array_equal(got, exp)
        assert ctx.stats().n_candidates > img.size // 2
        assert bool(ctx.stats().path_flags & PATH_STAGE_FALLBACK) == bool(fused)


@pytest.mark.parametrize("params", [dict(maxCorners=1500, q=0.05, md=7, bs=15), dict(maxCorners=0, q=0.2, md=3, bs=3),
                                    dict(maxCorners=400, q=0.01, md=12.5, bs=7), dict(maxCorners=300, q=0.1, md=5, bs=4)])
@pytest.mark.parametrize("shape", [(300, 700), (131, 267)])
def test_good_features_fused_kernel_bit_exact(ops, O, params, shape):
    """The optional fused eig+candidate kernel (km_set_option "fused_eig") gives the oracle's corners too: border strips
    (image narrower than one strip / W % 4 != 0), masks, even block sizes (those fall back to the two-kernel path)."""
    from karios_amd._lib import default_context
    lap_mon, lap_ref, mon, ref = _lap_pair(O, shape[0], shape[1], nodata_wedge=True)
    mask, _ = O.auto_mask(mon, ref)
    ctx = default_context()
    try:
        for fused in (1, 0):
            ctx.set_option("fused_eig", fused)
            for mk in (None, mask):
                got = ops.good_features_to_track(lap_ref, params["maxCorners"], params["q"], params["md"], mask=mk, blockSize=params["bs"])
                exp = O.good_features(lap_ref, mk, params["maxCorners"], params["q"], params["md"], params["bs"])
                if exp is None:
                    assert got is None
                else:
                    np.testing.assert_array_equal(got, exp)
    finally:
        ctx.set_option("fused_eig", 1)


def test_good_features_flat_image_is_none(ops):
    assert ops.good_features_to_track(np.full((80, 80), 17, np.uint8), 100, 0.1, 10, blockSize=15) is None
    assert ops.good_features_to_track(np.zeros((2, 2), np.uint8), 100, 0.1, 10, blockSize=3) is None


@pytest.fixture(params=[1, 0], ids=["lk_resident_patches", "lk_first_form"])
def lk_form(request, ops):
    """Both forms of the LK kernel (k_lk.hip): four resident patches per key point (default for two-level pyramids) and the
    first form (per-level staging + derivative planes; deeper pyramids, row bands)."""
    ctx = ops._lib.default_context()
    ctx.set_option("lk2", request.param)
    yield request.param
    ctx.set_option("lk2", 1)


@pytest.mark.parametrize("win", [25, 9, 31, 15, 21, 38])
def test_pyrlk_matches_oracle(ops, O, win, lk_form):
    lap_mon, lap_ref, _, _ = _lap_pair(O, 260, 330, sx=0.6, sy=-0.35)
    p0 = O.good_features(lap_ref, None, 800, 0.05, 7, 9)
    got = ops.calc_optical_flow_pyr_lk(lap_ref, lap_mon, p0, winSize=(win, win))
    exp = O.pyr_lk(lap_ref, lap_mon, p0, win)
    assert np.abs(got - exp).max() <= 1e-3
    assert np.array_equal(got, exp), f"not bit-identical: max diff {np.abs(got - exp).max()}"
    # backward pass from sub-pixel positions, including points near / beyond the border
    extra = np.array([[[0.3, 0.2]], [[329.2, 259.9]], [[-5.5, 10.0]], [[100.25, -30.0]], [[400.0, 400.0]]], np.float32)
    p1 = np.concatenate([got, extra])
    gb = ops.calc_optical_flow_pyr_lk(lap_mon, lap_ref, p1, winSize=(win, win))
    eb = O.pyr_lk(lap_mon, lap_ref, p1, win)
    assert np.abs(gb - eb).max() <= 1e-3
    assert np.array_equal(gb, eb)


@pytest.mark.parametrize("shift", [(0.5, 0.25), (2.6, -1.7), (4.4, 5.3), (-7.2, 3.1)])
def test_lk_forward_backward_in_one_launch_with_displacements_beyond_the_patch_margin(ops, O, shift, lk_form):
    """Forward + backward pass of the fused tracker launch (`km_klt_track` with given corners) against the oracle's two calls:
    displacements within the 3-px margin of the resident patches, and beyond it per level (patches re-centred for the search and
    for the backward template), plus corners whose windows overhang or leave the image."""
    lap_mon, lap_ref, _, _ = _lap_pair(O, 300, 360, sx=shift[0], sy=shift[1], seed=20260107)
    p0 = O.good_features(lap_ref, None, 600, 0.05, 7, 9)
    extra = np.array([[[0.0, 0.0]], [[359.0, 299.0]], [[3.0, 150.0]], [[356.0, 7.0]], [[180.0, 297.0]], [[12.0, 12.0]], [[-20.0, 40.0]],
                      [[500.0, 100.0]], [[100.0, -13.0]], [[13.0, 286.0]]], np.float32)
    p0 = np.concatenate([p0, extra])
    conf = O.default_conf(maxCorners=len(p0))
    got = ops.klt_track(lap_ref, lap_mon, None, conf, p0=p0)
    p1 = O.pyr_lk(lap_ref, lap_mon, p0, 25)
    p0r = O.pyr_lk(lap_mon, lap_ref, p1, 25)
    np.testing.assert_array_equal(got[0], p0)
    np.testing.assert_array_equal(got[1], p1)
    np.testing.assert_array_equal(got[2], p0r)
    moved = np.abs(p1 - p0).reshape(-1, 2).max(1)
    if max(abs(shift[0]), abs(shift[1])) > 4:
        assert (moved > 3).mean() > 0.5          # (the case really leaves the margin)


def test_lk_many_points_with_points_outside_the_image(ops, O):
    """>= 2048 key points in one launch (every XCD takes one contiguous eighth of the list; results are written by index), four of
    them outside the image (calcOpticalFlowPyrLK returns their start positions, App. A.3): equal to the oracle's."""
    lap_mon, lap_ref, _, _ = _lap_pair(O, 420, 520, sx=0.8, sy=-0.6, seed=20260109)
    yy, xx = np.mgrid[3:417:6, 2:518:6]
    p0 = np.stack([xx.ravel(), yy.ravel()], 1).astype(np.float32)
    p0 = np.concatenate([p0, np.array([[-30.0, 5.0], [600.0, 100.0], [100.0, 500.0], [-1e6, 1e6]], np.float32)]).reshape(-1, 1, 2)
    assert len(p0) >= 2048
    conf = O.default_conf(maxCorners=len(p0))
    p1 = O.pyr_lk(lap_ref, lap_mon, p0, 25)
    p0r = O.pyr_lk(lap_mon, lap_ref, p1, 25)
    got = ops.klt_track(lap_ref, lap_mon, None, conf, p0=p0)
    np.testing.assert_array_equal(got[1], p1)
    np.testing.assert_array_equal(got[2], p0r)


def test_pyrlk_small_image_no_pyramid_and_identity(ops, O):
    img = rand_u8((40, 44), seed=77)  # (w+1)/2 <= winSize: level 1 is dropped
    pts = np.array([[[10.0, 12.0]], [[30.5, 20.25]], [[0.0, 0.0]], [[43.0, 39.0]]], np.float32)
    got = ops.calc_optical_flow_pyr_lk(img, img, pts)
    np.testing.assert_array_equal(got, O.pyr_lk(img, img, pts))
    np.testing.assert_array_equal(got[:2], pts[:2])  # identical images: zero flow, exactly


def test_klt_track_and_tile_match_oracle(ops, O):
    mon, ref = synth.make_pair(384, 448, 0.5, 0.25)
    conf = O.default_conf(maxCorners=3000)
    exp = O.klt_tile(mon, ref, conf)
    # unfused: GPU tracker on the oracle's Laplacians
    tr = ops.klt_track(exp["lap_ref"], exp["lap_mon"], exp["mask"], conf)
    p0e = O.good_features(exp["lap_ref"], exp["mask"], conf.maxCorners, conf.qualityLevel, conf.minDistance, conf.blocksize)
    np.testing.assert_array_equal(tr[0], p0e)
    p1e = O.pyr_lk(exp["lap_ref"], exp["lap_mon"], p0e, 25)
    p0re = O.pyr_lk(exp["lap_mon"], exp["lap_ref"], p1e, 25)
    assert np.abs(tr[1] - p1e).max() <= 1e-3 and np.abs(tr[2] - p0re).max() <= 1e-3
    # fused tile pipeline from raw uint16
    status, tracks = ops.klt_tile(ref, mon, conf, mon_ksize=7, ref_ksize=7)
    assert status == "ok"
    np.testing.assert_array_equal(tracks[0], p0e)
    assert np.abs(tracks[1] - p1e).max() <= 1e-3 and np.abs(tracks[2] - p0re).max() <= 1e-3
    st = ops._lib.default_context().stats()
    assert st.valid_pixels == int(exp["mask"].sum()) and st.n_init == len(p0e)


def test_klt_tile_mixed_ksize_invert_mask_nodata(ops, O):
    mon, ref = synth.make_pair(300, 300, -0.4, 0.3, nodata_wedge=True)
    conf = O.default_conf(maxCorners=1500, laplacian_kernel_size={"mon": 5, "ref": 9})
    exp = O.klt_tile(mon, ref, conf, nodata_mon=1, nodata_ref=None, invert_mon=True)
    status, tracks = ops.klt_tile(ref, mon, conf, nodata_mon=1, mon_ksize=5, ref_ksize=9, invert_mon=True)
    assert status == "ok"
    p0e = O.good_features(exp["lap_ref"], exp["mask"], 1500, 0.1, 10, 15)
    np.testing.assert_array_equal(tracks[0], p0e)
    p1e = O.pyr_lk(exp["lap_ref"], exp["lap_mon"], p0e, 25)
    assert np.abs(tracks[1] - p1e).max() <= 1e-3


def test_klt_tile_no_valid_pixels(ops):
    z = np.zeros((64, 64), np.uint16)
    conf = type("C", (), dict(maxCorners=100, blocksize=15, matching_winsize=25, qualityLevel=0.1, minDistance=10))()
    assert ops.klt_tile(z, z, conf, mon_ksize=7, ref_ksize=7)[0] == "no_valid_pixels"
    flat = np.full((64, 64), 500, np.uint16)
    assert ops.klt_tile(flat, flat, conf, mon_ksize=7, ref_ksize=7)[0] == "no_features"


@pytest.mark.parametrize("dtype", [np.uint16, np.uint8, np.int16, np.float32])
def test_zncc_matches_oracle(ops, O, dtype):
    mon, ref = synth.make_pair(200, 240, 1.5, -2.5)
    mon, ref = mon.astype(dtype), ref.astype(dtype)
    rng = np.random.default_rng(5)
    n = 500
    x0 = rng.integers(-5, 245, n).astype(np.float32)
    y0 = rng.integers(-5, 205, n).astype(np.float32)
    dx = (rng.standard_normal(n) * 3).astype(np.float32)
    dy = (rng.standard_normal(n) * 3).astype(np.float32)
    dx[:20] = np.array([0.5, 1.5, 2.5, -0.5, -1.5] * 4, np.float32)  # half-to-even rounding cases
    got, exp = ops.zncc_batch(ref, mon, x0, y0, dx, dy), O.zncc_batch(ref, mon, x0, y0, dx, dy)
    assert np.array_equal(np.isnan(got), np.isnan(exp))
    assert np.nanmax(np.abs(got - exp)) <= 1e-9
    assert np.isnan(exp).sum() > 10 and (~np.isnan(exp)).sum() > 100


def test_zncc_zero_std_is_nan(ops):
    flat = np.full((100, 100), 100, np.uint16)
    z = np.array([50], np.float32)
    assert np.isnan(ops.zncc_batch(flat, flat, z, z, z * 0, z * 0)[0])


@pytest.mark.parametrize("shape,shift", [((64, 64), (7, -12)), ((96, 130), (-20, 31)), ((61, 45), (3, 5)), ((128, 128), (0, 0))])
def test_phase_correlation_integer_shift(ops, O, shape, shift):
    _, ref = synth.make_pair(shape[0], shape[1], 0, 0)
    mon = np.roll(ref, shift, (0, 1))
    got = ops.phase_cross_correlation(mon, ref)
    exp = O.phase_cross_correlation(mon, ref)
    np.testing.assert_array_equal(got, exp)
    np.testing.assert_array_equal(got, np.array(shift, np.float64))


@pytest.mark.parametrize("dtype", [np.uint16, np.uint8, np.int16])
@pytest.mark.parametrize("pad", [0, 1, 2, 4])
def test_phase_correlation_on_row_strided_rasters_in_both_precisions(ops, O, dtype, pad):
    """Rows of a wider array (stride != width; odd strides / offsets put the rows off the dword grid): the transforms' first level
    reads 8- / 16-bit pixels as whole dwords where every row start allows it and one by one otherwise - same answer either way, in
    float32 and in complex128, for a side with the 61-point level (61 * 12 = 732) and a smooth one."""
    from karios_amd._lib import default_context
    ctx = default_context()
    for (H, W), shift in (((96, 732), (-9, 21)), ((80, 120), (5, -7)), ((64, 121), (4, 9)), ((61, 122), (-3, 6)), ((40, 183), (2, -5))):   # widths 0 / 1 / 2 / 3 mod 4
        _, ref = synth.make_pair(H, W + 8, 0, 0, seed=H + W + pad)
        ref = ref[:, 2:2 + W + 4]
        if dtype is np.uint8:
            ref = (ref >> 5).clip(0, 255).astype(np.uint8)
        elif dtype is np.int16:
            ref = (ref.astype(np.int32) - 4000).astype(np.int16)
        wide_r = np.zeros((H, W + 4 + pad), dtype); wide_m = np.zeros((H, W + 4 + pad), dtype)
        off = pad % 3
        wide_r[:, off:off + W] = ref[:, :W]
        wide_m[:, off:off + W] = np.roll(ref[:, :W], shift, (0, 1))
        a, b = wide_m[:, off:off + W], wide_r[:, off:off + W]          # views: row stride W + 4 + pad, first pixel at `off`
        exp = O.phase_cross_correlation(np.ascontiguousarray(a), np.ascontiguousarray(b))
        np.testing.assert_array_equal(exp, np.array(shift, np.float64))
        for fp64 in (0, 1):
            ctx.set_option("phase_fp64", fp64)
            try:
                np.testing.assert_array_equal(ops.phase_cross_correlation(a, b), exp)
            finally:
                ctx.set_option("phase_fp64", 0)


def test_shift_image_matches_reference_semantics(ops, O):
    rng = np.random.default_rng(9)
    for dtype in (np.uint8, np.uint16, np.float32, np.float64):
        img = rng.integers(0, 200, (37, 53)).astype(dtype)
        for yo, xo in [(0, 0), (3, 0), (0, -4), (-5, 7), (2.6, -1.4), (40, 1), (1, -60)]:
            np.testing.assert_array_equal(ops.shift_image(img, yo, xo), O.shift_image(img, yo, xo))


def test_resident_pair_device_frame_equals_host_frame(ops, O):
    """ResidentPair.match_tile (FB test / score / ordering on the device, K8) == the reference-style numpy/pandas
    assembly of the same tracks, row for row including the permuted index labels; tiles of a resident image."""
    import pandas as pd
    from karios_amd.core import KLTConfiguration
    from karios_amd import frames as F
    from karios_amd.matcher.klt import KLT
    from karios_amd.resident import ResidentPair
    mon, ref = synth.make_pair(420, 500, 0.5, 0.25)
    conf = KLTConfiguration(tile_size=260, maxCorners=700)
    pair = ResidentPair.upload(mon, ref)
    frames = list(pair.match(conf))
    boxes = KLT(conf).tile_boxes(500, 420)
    assert len(frames) == len(boxes) == 4
    for f, (xo, yo, bx, by) in zip(frames, boxes):
        status, tracks = pair.track_tile(conf, (xo, yo, bx, by))
        exp = F.assemble(F.track_columns(*tracks)[0], x_off=xo, y_off=yo)
        pd.testing.assert_frame_equal(f, exp, check_exact=True)
        o = O.klt_tile(mon[yo:yo + by, xo:xo + bx], ref[yo:yo + by, xo:xo + bx], conf, x_off=xo, y_off=yo)
        np.testing.assert_array_equal(f["x0"].to_numpy(), o["x0"])
        np.testing.assert_array_equal(f["score"].to_numpy(), o["score"])
    fused = pair.match_tile(conf, boxes[0], zncc_threshold=0.4)
    scored = pair.score_frame(frames[0].copy(), 0.4)
    np.testing.assert_array_equal(fused["zncc_score"].to_numpy(), scored["zncc_score"].to_numpy())
    assert list(pair.score_frame(fused, 0.4).columns) == ["x0", "y0", "dx", "dy", "score", "zncc_score", "radial error", "angle"]
    z = O.zncc_batch(ref, mon, scored["x0"].to_numpy(), scored["y0"].to_numpy(), scored["dx"].to_numpy(), scored["dy"].to_numpy())
    keep = scored["score"].to_numpy() >= 0.4
    got = scored["zncc_score"].to_numpy()
    assert np.all(np.isnan(got[~keep])) and np.nanmax(np.abs(got[keep] - z[keep])) <= 1e-9
    full = pair.score_frame(pair.match_tile(conf, boxes[0], zncc_threshold=0.4), 0.4, mutual_info=True)
    assert list(full.columns)[-2:] == ["mutual_info_score", "mi_score"]
    est, enmi = O.mi_batch(ref, mon, *(full[c].to_numpy()[keep] for c in ("x0", "y0", "dx", "dy")))
    assert np.array_equal(np.isnan(full["mi_score"].to_numpy()[keep]), np.isnan(enmi))
    assert np.nanmax(np.abs(full["mutual_info_score"].to_numpy()[keep] - est)) <= 1e-9 and np.nanmax(np.abs(full["mi_score"].to_numpy()[keep] - enmi)) <= 1e-9
    assert np.all(np.isnan(full["mi_score"].to_numpy()[~keep]))
    flat = ResidentPair.upload(np.full((64, 64), 7, np.uint16), np.full((64, 64), 7, np.uint16))
    assert flat.match_tile(KLTConfiguration()) is None


def test_resident_pair_pipelined_match_equals_tile_by_tile(ops, O):
    """`match_pipelined` (host half of tile i overlapped with the device half of tile i+1) yields the frames of `match`."""
    import pandas as pd
    from karios_amd.core import KLTConfiguration
    from karios_amd.resident import ResidentPair
    mon, ref = synth.make_pair(400, 620, 0.4, -0.3, seed=5)
    conf = KLTConfiguration(maxCorners=600, tile_size=200, xStart=10)
    pair = ResidentPair.upload(mon, ref)
    want = list(pair.match(conf))
    got = list(pair.match_pipelined(conf))
    assert len(want) == len(got) > 2
    for a, b in zip(want, got):
        pd.testing.assert_frame_equal(a, b)
    scored = list(pair.match_pipelined(conf, zncc_threshold=0.4, host_stage=lambda f: pair.score_frame(f, 0.4)))
    for a, b in zip(want, scored):
        ref_frame = pair.score_frame(a.copy(), 0.4)
        pd.testing.assert_frame_equal(ref_frame[sorted(ref_frame.columns)], b[sorted(b.columns)])


def test_submit_wait_frames_in_flight(ops, O):
    """km_klt_tile_frame_submit / km_frame_wait: three tiles in flight, waited for out of order and from another thread,
    give the blocks of the synchronous call; an unwaited slot is recycled; a stale ticket is refused."""
    import threading
    import pandas as pd
    from karios_amd._lib import KariosHipError
    from karios_amd.core import KLTConfiguration
    from karios_amd.resident import ResidentPair
    mon, ref = synth.make_pair(420, 610, -0.3, 0.45, seed=11, nodata_wedge=True)
    conf = KLTConfiguration(maxCorners=700)
    pair = ResidentPair.upload(mon, ref)
    boxes = [(0, 0, 300, 200), (300, 0, 310, 420), (5, 200, 280, 220), None]
    want = [pair.match_tile(conf, box=b, zncc_threshold=0.4) for b in boxes]
    pair.ctx.set_profiling(True)
    # this test is about the slot ring: with "speculative" on, a tile that does not fit the sync-free corner path comes back
    # flagged and `PendingFrame.result()` repeats it (tests/test_gpu_forced_paths.py); here every frame must be final on wait()
    pair.ctx.set_option("speculative", 0)
    try:
        pend = [pair.submit_tile(conf, box=b, zncc_threshold=0.4) for b in boxes[:3]]
        got = [None] * 3
        th = threading.Thread(target=lambda: got.__setitem__(2, pend[2].wait().to_frame()))
        th.start()
        got[1] = pend[1].wait().to_frame()
        got[0] = pend[0].wait().to_frame()
        th.join()
        spans = pend[1].stage_ms()
        assert spans["lk_fwd_bwd"] > 0 and spans["stretch_laplacian_mask"] > 0 and spans["zncc"] > 0
    finally:
        pair.ctx.set_profiling(False)
    for a, b in zip(want[:3], got):
        pd.testing.assert_frame_equal(a, b)
    try:
        _rest_of_submit_wait_test(pair, conf, boxes, want)
    finally:
        pair.ctx.set_option("speculative", 0)


@pytest.mark.parametrize("mutual_info", [False, True])
def test_submitted_units_of_different_boxes_two_in_flight(ops, O, mutual_info):
    """Twelve submitted units of four different tile boxes back to back, two in flight, equal the blocking call's frames - with the
    sync-free corner path, then the exact path, and a blocking call in between (klt.py:220-253: the per-tile loop, pipelined)."""
    import pandas as pd
    from karios_amd.core import KLTConfiguration
    from karios_amd.resident import ResidentPair
    mon, ref = synth.make_pair(700, 900, 0.35, -0.4, seed=23, nodata_wedge=True)
    conf = KLTConfiguration(maxCorners=900)
    pair = ResidentPair.upload(mon, ref)
    boxes = [(0, 0, 450, 350), (450, 0, 450, 700), (10, 350, 400, 330), None]
    want = [pair.match_tile(conf, box=b, zncc_threshold=0.4, mutual_info=mutual_info) for b in boxes]
    try:
        for spec in (1, 0):
            pair.ctx.set_option("speculative", spec)
            pend = []
            for i in range(12):
                pend.append((i % 4, pair.submit_tile(conf, box=boxes[i % 4], zncc_threshold=0.4, mutual_info=mutual_info)))
                if len(pend) == 2:
                    k, p = pend.pop(0)
                    pd.testing.assert_frame_equal(want[k], p.result().to_frame())
                if i == 6:                                   # a blocking call in between waits for the tail in flight
                    pd.testing.assert_frame_equal(want[2], pair.match_tile(conf, box=boxes[2], zncc_threshold=0.4, mutual_info=mutual_info))
            for k, p in pend:
                pd.testing.assert_frame_equal(want[k], p.result().to_frame())
    finally:
        pair.ctx.set_option("speculative", 1)


def _rest_of_submit_wait_test(pair, conf, boxes, want):
    import pandas as pd
    from karios_amd._lib import KariosHipError
    from karios_amd.resident import ResidentPair
    # four submissions without a wait: the first slot is recycled (its frame is lost), the last three are intact
    pend = [pair.submit_tile(conf, box=b, zncc_threshold=0.4) for b in boxes]
    for a, p in zip(want[1:], pend[1:]):
        pd.testing.assert_frame_equal(a, p.wait().to_frame())
    with pytest.raises(KariosHipError):
        pend[1].__class__(pair.ctx, pend[1].ticket, pend[1].cap, True).wait()   # that slot has been consumed
    # no ZNCC column, no-features tile
    p = pair.submit_tile(conf, box=boxes[0])
    pd.testing.assert_frame_equal(pair.match_tile(conf, box=boxes[0]), p.wait().to_frame())
    flat = ResidentPair.upload(np.full((64, 64), 7, np.uint16), np.full((64, 64), 7, np.uint16))
    assert flat.submit_tile(conf).wait().to_frame() is None


def test_auto_ksize_batched_search_matches_oracle_loop(ops, O):
    """km_klt_auto_ksize_frame_dev == the reference's 5x5 loop (klt.py:465-545) done with the oracle:
    every inlier ratio, the winning pair (first maximum in itertools.product order) and the winner's frame."""
    import itertools
    from karios_amd.core import KLTConfiguration
    from karios_amd.resident import ResidentPair
    mon, ref = synth.make_cross_sensor_pair(236, 300, seed=9)[:2]
    mask = np.ones(ref.shape, np.uint8)
    mask[:40, :50] = 0
    conf = KLTConfiguration(maxCorners=500, minDistance=6, blocksize=7, laplacian_kernel_size="auto")
    cands = [3, 5, 7, 9, 11]
    laps_m = {k: O.laplacian_u8(O.to_uint8(mon), k) for k in cands}
    laps_r = {k: O.laplacian_u8(O.to_uint8(ref), k) for k in cands}
    p0s = {k: O.good_features(laps_r[k], mask, conf.maxCorners, conf.qualityLevel, conf.minDistance, conf.blocksize) for k in cands}
    want, best, best_ratio, best_res = {}, None, -1.0, None
    for mk, rk in itertools.product(cands, repeat=2):
        res = None if p0s[rk] is None else O.klt_tracker(laps_r[rk], laps_m[mk], mask, conf, p0=p0s[rk])
        if res is None:
            want[(mk, rk)] = 0.0
            continue
        pts, ninit = res
        want[(mk, rk)] = len(pts["x0"]) / ninit if ninit else 0.0
        if want[(mk, rk)] > best_ratio:
            best_ratio, best, best_res = want[(mk, rk)], (mk, rk), res
    pair = ResidentPair.upload(mon, ref, mask)
    frame, scores, got_best, ninit = pair.match_tile_auto_ksize(conf, candidates=cands)
    assert scores == pytest.approx(want, abs=0) and got_best == best and ninit == best_res[1]
    pts = best_res[0]
    order = np.lexsort((pts["y0"], pts["x0"]))
    for col in ("x0", "y0", "dx", "dy", "score"):
        assert np.array_equal(frame[col].to_numpy(), np.asarray(pts[col], np.float32)[order]), col


def test_error_contract_like_cv2(ops):
    """Malformed inputs raise (the reference gets cv2.error / ValueError), they never return garbage."""
    img = rand_u8((40, 40))
    E = ops.KariosHipError
    with pytest.raises(E):
        ops.to_uint8(np.zeros((4, 4), np.float64))                       # unsupported pixel type
    with pytest.raises(E):
        ops.calc_optical_flow_pyr_lk(img, img[:, :30], np.zeros((1, 1, 2), np.float32))
    with pytest.raises(E):
        ops.calc_optical_flow_pyr_lk(img, img, np.zeros((1, 1, 2), np.float32), winSize=(2, 2))
    with pytest.raises(E):
        ops.good_features_to_track(img, 10, 0.0, 5)                      # qualityLevel must be > 0
    with pytest.raises(E):
        ops.good_features_to_track(img, 10, 0.1, -1)
    with pytest.raises(E):
        ops.min_eigen(img, 0)
    with pytest.raises(E):
        ops.phase_cross_correlation(img, img[:20])
    with pytest.raises(E):
        ops.zncc_batch(img, img.astype(np.uint16), [1], [1], [0], [0])


def test_two_threads_two_contexts(ops, O):
    """klt_tracker is called from a ThreadPoolExecutor in the reference's auto-ksize mode (klt.py:526-527): every
    thread gets its own context (stream + workspace) and results do not depend on the interleaving."""
    from concurrent.futures import ThreadPoolExecutor
    from karios_amd.matcher import klt_tracker
    mon, ref = synth.make_pair(220, 260, 0.4, 0.1)
    lm, lr = O.laplacian_u8(O.to_uint8(mon), 5), O.laplacian_u8(O.to_uint8(ref), 5)
    conf = O.default_conf(maxCorners=400)
    exp = klt_tracker(lr, lm, None, conf)[0]
    with ThreadPoolExecutor(4) as ex:
        outs = list(ex.map(lambda _: klt_tracker(lr, lm, None, conf)[0], range(12)))
    assert all(o.equals(exp) for o in outs)


def test_large_offset_flow_and_gdt_byte_quirk(ops):
    from karios_amd.core import NumpyRasterImage
    from karios_amd.matcher import detect_large_offset
    _, ref = synth.make_pair(128, 160, 0, 0)
    mon = np.roll(ref, (5, -9), (0, 1))
    res = detect_large_offset(NumpyRasterImage(ref), NumpyRasterImage(mon))
    shifted, x_off, y_off = res
    assert (x_off, y_off) == (-9.0, 5.0) and shifted.dtype == np.uint16
    np.testing.assert_array_equal(shifted[:-5, 9:], ref[:-5, 9:])   # content moved back onto the reference grid
    q = detect_large_offset(NumpyRasterImage(ref), NumpyRasterImage(mon), emulate_gdt_byte_write=True)[0]
    assert q.dtype == np.uint8 and q.max() == 255
    small = np.roll(ref, (1, 1), (0, 1))                                # below bias_correction_min_threshold on both axes
    assert detect_large_offset(NumpyRasterImage(ref), NumpyRasterImage(small)) is None


@pytest.mark.parametrize("shape,seed", [((97, 131), 1), ((256, 256), 2), ((333, 517), 3), ((64, 1025), 4)])
def test_user_mask_valid_pixel_count(ops, shape, seed):
    """klt.py:266-279: with a user mask the valid pixels are its non-zero bytes (any value, not only 255) - counted 16 bytes at a
    time on the device; sizes that leave unaligned heads and tails, masks that are all zero / all set."""
    rng = np.random.default_rng(seed)
    mon, ref = synth.make_pair(shape[0], shape[1], 0.3, -0.2, seed=seed)
    conf = __import__("karios_amd.core", fromlist=["KLTConfiguration"]).KLTConfiguration(maxCorners=200)
    for kind in ("random", "sparse", "zero", "full"):
        mask = {"random": rng.integers(0, 256, shape).astype(np.uint8) * (rng.random(shape) < 0.7),
                "sparse": ((rng.random(shape) < 0.01) * rng.integers(1, 256, shape)).astype(np.uint8),
                "zero": np.zeros(shape, np.uint8), "full": np.full(shape, 255, np.uint8)}[kind].astype(np.uint8)
        status, _ = ops.klt_tile(ref, mon, conf, mask_box=mask, mon_ksize=7, ref_ksize=7)
        st = ops._lib.default_context().stats()
        if kind == "zero":
            assert status == "no_valid_pixels"
        else:
            assert st.valid_pixels == int(np.count_nonzero(mask)), (kind, shape)


def test_lk_oscillation_literal(ops, O):
    """The LK kernels' oscillation stop == the oracle's == OpenCV's declared types: float32 |delta + prevDelta| against the DOUBLE
    literal 0.01, driven to exactly float32(0.01) and one ulp either side."""
    from test_oracle_golden import _oscillation_quads
    q, want = _oscillation_quads()
    np.testing.assert_array_equal(ops.lk_oscillation_probe(q), want)
    rng = np.random.default_rng(9)
    r = rng.uniform(-0.02, 0.02, (4000, 4)).astype(np.float32)
    np.testing.assert_array_equal(ops.lk_oscillation_probe(r), np.array([O.lk_oscillates(*row) for row in r]))


@pytest.mark.parametrize("dtype", [np.int16, np.uint16, np.float32, np.uint8])
def test_klt_tile_from_pageable_strided_box_through_the_ring(ops, O, dtype, monkeypatch):
    """klt.py:252-253: a tile is matched as read.  Host rasters are pageable numpy views (row stride > width, odd offsets); they
    travel through the library's page-locked staging ring (csrc/staging.hip) in several chunks - the ring is shrunk to 64-KB
    slots so that a 700 x 900 box wraps it many times - and the result equals the oracle's on the packed box."""
    import subprocess, sys, os, textwrap
    code = textwrap.dedent(f"""
        import numpy as np, sys
        sys.path.insert(0, {os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r})
        from karios_amd import ops, synth
        from oracle import oracle as O
        mon, ref = synth.make_pair(900, 1100, 0.4, -0.3, seed=77)
        dt = np.dtype({np.dtype(dtype).name!r})
        if dt == np.uint8:
            mon, ref = (mon >> 5).astype(np.uint8), (ref >> 5).astype(np.uint8)
        elif dt == np.int16:
            mon, ref = (mon.astype(np.int32) - 4000).astype(np.int16), (ref.astype(np.int32) - 4000).astype(np.int16)
        elif dt == np.float32:
            mon, ref = mon.astype(np.float32) * np.float32(0.37), ref.astype(np.float32) * np.float32(0.37)
        mb, rb = mon[101:801, 57:957], ref[101:801, 57:957]
        conf = O.default_conf(maxCorners=2000)
        for rep in range(3):
            st, tr = ops.klt_tile(rb, mb, conf, mon_ksize=7, ref_ksize=7)
            assert st == "ok"
            exp = O.klt_tile(np.ascontiguousarray(mb), np.ascontiguousarray(rb), conf)
            p0 = O.good_features(exp["lap_ref"], exp["mask"], conf.maxCorners, conf.qualityLevel, conf.minDistance, conf.blocksize)
            p1 = O.pyr_lk(exp["lap_ref"], exp["lap_mon"], p0, 25)
            p0r = O.pyr_lk(exp["lap_mon"], exp["lap_ref"], p1, 25)
            assert np.array_equal(tr[0], p0) and np.array_equal(tr[1], p1) and np.array_equal(tr[2], p0r)
        from karios_amd._lib import default_context
        armed, missed = default_context().upload_check_stats()
        assert armed >= 6 and missed == 0, (armed, missed)
        print("ring ok", armed)
    """)
    env = dict(os.environ, KARIOS_HIP_RING_CHUNK_KB="64", KARIOS_HIP_UPLOAD_CHECKSUM="1")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "ring ok" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
    assert "MISS" not in out.stderr


def test_large_results_leave_through_the_landing_arena(ops, O):
    """Results larger than the page-locked landing arena (8 MB) are copied out piecewise: a 3000 x 3100 min-eigenvalue map (37 MB)
    and its uint8 stretch equal the oracle's."""
    rng = np.random.default_rng(11)
    a = rng.integers(0, 256, (3000, 3100), dtype=np.uint8)
    np.testing.assert_array_equal(ops.laplacian_u8(a, 5), O.laplacian_u8(a, 5))
    b = rng.integers(0, 9000, (2100, 2300)).astype(np.uint16)
    np.testing.assert_array_equal(ops.to_uint8(b), O.to_uint8(b))
    e = ops.min_eigen(a[:1500, :1700], 5)
    np.testing.assert_array_equal(e, O.min_eigen(np.ascontiguousarray(a[:1500, :1700]), 5))
