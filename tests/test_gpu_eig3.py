"""GPU parity of the 8-pixels-per-lane fused minimum-eigenvalue + candidate kernel (karios_amd/csrc/k_eig3.hip), which serves
goodFeaturesToTrack (reference call site karios/matcher/klt.py:120) on images at least 512 columns wide: corners bit-identical
to the oracle's and to the 2-px kernel's on strips that end anywhere (shifted last strip, 12-column border strips run by the
2-px item), short / tall items, masks, every odd block size, exact ties (several candidates in one lane) and a forced stage
overflow."""
import numpy as np
import pytest

from karios_amd import synth

pytestmark = pytest.mark.gpu


def _lap(O, H, W, k=7, seed=5, **kw):
    mon, ref = synth.make_pair(H, W, 0.5, 0.25, seed=seed, **kw)
    return O.laplacian_u8(O.to_uint8(ref), k), mon, ref


@pytest.fixture
def ctx():
    from karios_amd._lib import default_context
    c = default_context()
    yield c
    c.set_option("eig3", 1)
    c.set_option("stage_cap", 0)


@pytest.mark.parametrize("shape", [(140, 512), (75, 523), (260, 1000), (47, 1511), (333, 1047), (1200, 640)])
@pytest.mark.parametrize("params", [dict(n=0, q=0.02, md=0, bs=15), dict(n=3000, q=0.1, md=10, bs=15), dict(n=0, q=0.1, md=2, bs=3),
                                    dict(n=900, q=0.05, md=6.5, bs=7), dict(n=0, q=0.3, md=1, bs=1)])
def test_corners_match_oracle_and_the_two_pixel_kernel(ops, O, ctx, shape, params):
    lap, mon, ref = _lap(O, *shape, nodata_wedge=True)
    mask, _ = O.auto_mask(mon, ref)
    for mk in (None, mask):
        exp = O.good_features(lap, mk, params["n"], params["q"], params["md"], params["bs"])
        for eig3 in (1, 0):
            ctx.set_option("eig3", eig3)
            got = ops.good_features_to_track(lap, params["n"], params["q"], params["md"], mask=mk, blockSize=params["bs"])
            if exp is None:
                assert got is None
            else:
                np.testing.assert_array_equal(got, exp)


@pytest.mark.parametrize("bs", [5, 9, 11, 13])
def test_every_odd_block_size(ops, O, ctx, bs):
    lap, mon, ref = _lap(O, 180, 777, k=5, seed=bs)
    exp = O.good_features(lap, None, 0, 0.05, 3, bs)
    got = ops.good_features_to_track(lap, 0, 0.05, 3, blockSize=bs)
    np.testing.assert_array_equal(got, exp)


def test_ties_several_candidates_per_lane_and_stage_overflow(ops, O, ctx):
    from karios_amd._lib import PATH_STAGE_FALLBACK
    # periodic pattern: exact ties everywhere (every lane holds several candidates per row), order decided by the raster-index rule
    tile = np.zeros((16, 4), np.uint8)
    tile[4:9, 1:3] = 255
    img = np.tile(tile, (20, 200))                     # 320 x 800, period 4 in x: two candidates per lane and row
    rng = np.random.default_rng(3)
    img[100:140, 300:420] = rng.integers(0, 256, (40, 120))    # and a random patch that sets the maximum
    for params in (dict(n=0, q=0.0005, md=0, bs=3), dict(n=5000, q=0.001, md=3, bs=5)):
        exp = O.good_features(img, None, params["n"], params["q"], params["md"], params["bs"])
        got = ops.good_features_to_track(img, params["n"], params["q"], params["md"], blockSize=params["bs"])
        np.testing.assert_array_equal(got, exp)
    # a stage of 48 slots overflows on a dense image: the call falls back to map + candidate scan and still agrees
    lap, mon, ref = _lap(O, 300, 900, k=3)
    exp = O.good_features(lap, None, 0, 0.001, 0, 3)
    ctx.set_option("stage_cap", 48)
    got = ops.good_features_to_track(lap, 0, 0.001, 0, blockSize=3)
    np.testing.assert_array_equal(got, exp)
    assert ctx.stats().path_flags & PATH_STAGE_FALLBACK


def test_whole_tile_with_wide_image(ops, O, ctx):
    """The tile pipeline on a wide pair: identical frames with either kernel, equal to the oracle's corners and tracks."""
    from karios_amd.core import KLTConfiguration
    from karios_amd.resident import ResidentPair
    mon, ref = synth.make_pair(700, 1300, -0.4, 0.3, seed=9)
    conf = KLTConfiguration(maxCorners=4000)
    pair = ResidentPair.upload(mon, ref)
    frames = []
    for eig3 in (1, 0):
        ctx.set_option("eig3", eig3)
        frames.append(pair.match_tile(conf, zncc_threshold=0.4))
    assert len(frames[0]) == len(frames[1]) > 1000
    for c in ("x0", "y0", "dx", "dy", "score", "zncc_score"):
        np.testing.assert_array_equal(frames[0][c].to_numpy(), frames[1][c].to_numpy())
    exp = O.klt_tile(mon, ref, O.default_conf(maxCorners=4000))
    p0 = O.good_features(exp["lap_ref"], exp["mask"], 4000, 0.1, 10, 15)
    ctx.set_option("eig3", 1)
    status, tracks = ops.klt_tile(ref, mon, O.default_conf(maxCorners=4000), mon_ksize=7, ref_ksize=7)
    assert status == "ok"
    np.testing.assert_array_equal(tracks[0], p0)
