"""CPU suite, part 1: pin the oracle.

(a) numpy-defined pieces against golden vectors produced by the reference's own functions
    (tests/golden/make_golden.py);
(b) OpenCV-defined pieces ("parity unpinned": cv2 is not installable here) against known-answer tests
    derived from the published algorithm (SURVEY.md 8c list) and against independent scipy.ndimage
    restatements.
"""
import os

import numpy as np
import pytest
from scipy import ndimage as ndi

from conftest import rand_u8
from karios_amd import synth

G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return np.load(os.path.join(G, name))


# ------------------------------------------------------------------ (a) golden vectors from the reference
def test_to_uint8_golden(O):
    g = load("to_uint8.npz")
    for k in [k[3:] for k in g.files if k.startswith("in_")]:
        np.testing.assert_array_equal(O.to_uint8(g["in_" + k]), g["out_" + k], err_msg=k)


def test_shift_image_golden(O):
    g = load("shift_image.npz")
    for i, (yo, xo) in enumerate(g["offsets"]):
        np.testing.assert_array_equal(O.shift_image(g["img"], yo, xo), g[f"out_{i}"])


def test_filter_outliers_golden(O):
    g = load("outliers.npz")
    r = O.filter_outliers(g["x0"], g["y0"], g["x1"], g["y1"], g["score"])
    for i, v in enumerate(r):
        np.testing.assert_array_equal(v, g[f"out_{i}"])
    assert len(r[0]) < len(g["x0"])


def test_zncc_golden(O):
    g = load("zncc.npz")
    for ref_key, exp_key in (("ref", "zncc"), ("ref_flat", "zncc_flat")):
        got = O.zncc_batch(g[ref_key], g["mon"], g["x0"], g["y0"], g["dx"], g["dy"])
        exp = g[exp_key]
        assert np.array_equal(np.isnan(got), np.isnan(exp))
        assert np.nanmax(np.abs(got - exp)) <= 1e-9
    assert np.isnan(g["zncc_flat"]).sum() > np.isnan(g["zncc"]).sum() > 0
    assert bool(g["uniform_is_nan"])


@pytest.mark.parametrize("case", ["tiles", "xstart", "mixed_inv", "usermask"])
def test_klt_match_glue_golden(O, case):
    """The oracle's own tile composition (oracle.klt_tile) equals the reference's KLT.match run with the
    oracle behind cv2: tile order / offsets / mask / sort / polarity / per-image ksize glue is pinned."""
    from karios_amd.core import KLTConfiguration
    from karios_amd.matcher.klt import KLT
    g = load(f"klt_match_{case}.npz")
    confs = {
        "tiles": dict(tile_size=200, maxCorners=600, laplacian_kernel_size=7),
        "xstart": dict(tile_size=130, xStart=130, maxCorners=300, laplacian_kernel_size=5),
        "mixed_inv": dict(tile_size=20000, maxCorners=800, laplacian_kernel_size={"mon": 5, "ref": 9},
                          laplacian_invert_polarity=True, outliers_filtering=True),
        "usermask": dict(tile_size=20000, maxCorners=500, laplacian_kernel_size=3),
    }
    conf = KLTConfiguration(**confs[case])
    mon, ref = g["mon"], g["ref"]
    nd = [None if np.isnan(v) else v for v in g["nodata"]]
    mask = g["mask"] if "mask" in g.files else None
    frames = []
    for x_off, y_off, bx, by in KLT(conf).tile_boxes(mon.shape[1], mon.shape[0]):
        sl = (slice(y_off, y_off + by), slice(x_off, x_off + bx))
        r = O.klt_tile(mon[sl], ref[sl], conf, mask_box=None if mask is None else mask[sl], nodata_mon=nd[0], nodata_ref=nd[1],
                       x_off=x_off, y_off=y_off, invert_mon=bool(conf.laplacian_invert_polarity))
        if r is not None:
            frames.append(r)
    assert len(frames) == int(g["n_frames"])
    for i, f in enumerate(frames):
        for col in ("x0", "y0", "dx", "dy", "score"):
            np.testing.assert_array_equal(f[col], g[f"f{i}_{col}"], err_msg=f"{case} frame {i} {col}")


def test_phase_correlation_vs_skimage(O):
    g = load("phase_corr.npz")
    ref = g["ref"]
    for s, exp in zip(g["shifts"], g["skimage_shift"]):
        mon = np.roll(ref, tuple(s), (0, 1))
        got = O.phase_cross_correlation(mon, ref)
        np.testing.assert_array_equal(got, exp)
    np.testing.assert_array_equal(g["skimage_shift"], g["shifts"].astype(np.float64))


# ------------------------------------------------------------------ (b) OpenCV-defined pieces: KATs
def test_sobel_kernels():
    from oracle import oracle as O
    assert O.sobel_kernel(5, 2).tolist() == [1, 0, -2, 0, 1] and O.sobel_kernel(5, 0).tolist() == [1, 4, 6, 4, 1]
    assert O.sobel_kernel(7, 2).tolist() == [1, 2, -1, -4, -1, 2, 1] and O.sobel_kernel(7, 0).tolist() == [1, 6, 15, 20, 15, 6, 1]
    assert O.sobel_kernel(9, 2).tolist() == [1, 4, 4, -4, -10, -4, 4, 4, 1]
    assert O.sobel_kernel(11, 0).tolist() == [1, 10, 45, 120, 210, 252, 210, 120, 45, 10, 1]


@pytest.mark.parametrize("ksize", [1, 3, 5, 7, 9, 11])
def test_laplacian_vs_scipy(O, ksize):
    img = rand_u8((67, 93), seed=ksize)
    I = img.astype(np.int64)
    if ksize in (1, 3):
        K = np.array([[0, 1, 0], [1, -4, 1], [0, 1, 0]]) if ksize == 1 else np.array([[2, 0, 2], [0, -8, 0], [2, 0, 2]])
        full = ndi.correlate(I, K, mode="mirror")
    else:
        kd, ks = O.sobel_kernel(ksize, 2).astype(np.int64), O.sobel_kernel(ksize, 0).astype(np.int64)
        full = (ndi.correlate1d(ndi.correlate1d(I, kd, axis=1, mode="mirror"), ks, axis=0, mode="mirror")
                + ndi.correlate1d(ndi.correlate1d(I, ks, axis=1, mode="mirror"), kd, axis=0, mode="mirror"))
    np.testing.assert_array_equal(O.laplacian_u8(img, ksize), np.clip(full, 0, 255).astype(np.uint8))


def test_laplacian_11_is_the_binomial_of_the_unsaturated_laplacian_9(O):
    """What the marching kernel's radius-5 form rests on (k_dense.hip lap_march_item): OpenCV's size-11 derivative / smoothing kernels
    are the size-9 kernels convolved with [1 2 1], so Laplacian_11 = ([1 2 1] x [1 2 1]) * (kd9 x ks9 + ks9 x kd9) on the REFLECT_101
    extension, saturated once at the end - also for images smaller than the kernel (multiple reflections)."""
    kd9, ks9, kd11, ks11 = (O.sobel_kernel(9, 2), O.sobel_kernel(9, 0), O.sobel_kernel(11, 2), O.sobel_kernel(11, 0))
    assert np.array_equal(np.convolve(kd9, [1, 2, 1]), kd11) and np.array_equal(np.convolve(ks9, [1, 2, 1]), ks11)

    def sep(img, kx, ky):             # rows filtered with kx, columns with ky, REFLECT_101, int64
        def along(a, k, axis):
            r = len(k) // 2
            n = a.shape[axis]
            idx = np.arange(-r, n + r)
            per = max(1, 2 * n - 2)
            idx = np.abs((idx % per + per) % per)
            idx = np.where(idx >= n, per - idx, idx) if n > 1 else np.zeros_like(idx)
            ext = np.take(a, idx, axis=axis)
            out = np.zeros_like(a)
            for t, c in enumerate(k):
                out += int(c) * np.take(ext, np.arange(t, t + n), axis=axis)
            return out
        return along(along(img.astype(np.int64), kx, 1), ky, 0)

    for seed, shape in enumerate([(40, 53), (64, 64), (7, 9), (3, 30), (21, 5)]):
        img = rand_u8(shape, seed=300 + seed)
        l9 = sep(img, kd9, ks9) + sep(img, ks9, kd9)                                     # unsaturated
        composite = np.clip(sep(l9, [1, 2, 1], [1, 2, 1]), 0, 255).astype(np.uint8)
        np.testing.assert_array_equal(composite, O.laplacian_u8(img, 11), err_msg=str(shape))


def test_laplacian_kats(O):
    ramp = np.tile(np.arange(40, dtype=np.uint8) * 3, (30, 1))
    for k in (1, 3, 5, 7):
        assert O.laplacian_u8(ramp, k)[8:-8, 8:-8].max() == 0          # linear ramp -> zero interior
    imp = np.zeros((21, 21), np.uint8)
    imp[10, 10] = 1
    assert O.laplacian_u8(imp, 3)[9:12, 9:12].tolist() == [[2, 0, 2], [0, 0, 0], [2, 0, 2]]  # -8 saturates to 0
    imp[10, 10] = 255
    assert O.laplacian_u8(imp, 1)[9:12, 9:12].tolist() == [[0, 255, 0], [255, 0, 255], [0, 255, 0]]
    k7 = O.laplacian_u8((imp > 0).astype(np.uint8), 7)                 # impulse response of kd x ks + ks x kd
    kd, ks = O.sobel_kernel(7, 2), O.sobel_kernel(7, 0)
    exp = np.clip(np.outer(ks, kd) + np.outer(kd, ks), 0, 255)
    np.testing.assert_array_equal(k7[7:14, 7:14], exp[::-1, ::-1])
    with pytest.raises(ValueError):
        O.laplacian_u8(imp, 4)


def test_min_eigen_vs_scipy(O):
    img = O.laplacian_u8(O.to_uint8(synth.make_pair(90, 130)[1]), 3)
    I = img.astype(np.int64)
    sx = ndi.correlate1d(ndi.correlate1d(I, [-1, 0, 1], axis=1, mode="mirror"), [1, 2, 1], axis=0, mode="mirror")
    sy = ndi.correlate1d(ndi.correlate1d(I, [1, 2, 1], axis=1, mode="mirror"), [-1, 0, 1], axis=0, mode="mirror")
    for bs in (15, 3, 4):
        one = np.ones(bs, np.int64)
        # scipy centres an even window at index bs//2 = OpenCV's default anchor: [x - bs//2, x + bs - 1 - bs//2]
        box = lambda a: ndi.correlate1d(ndi.correlate1d(a, one, axis=1, mode="mirror"), one, axis=0, mode="mirror")
        sc = (1.0 / (4 * bs * 255)) ** 2
        cxx, cxy, cyy = ((box(p) * sc).astype(np.float32) for p in (sx * sx, sx * sy, sy * sy))
        a, c, b = cxx * np.float32(.5), cyy * np.float32(.5), cxy
        exp = (a + c) - np.sqrt((a - c) * (a - c) + b * b)
        np.testing.assert_array_equal(O.min_eigen(img, bs), exp, err_msg=f"block {bs}")


def _blob_image():
    img = np.zeros((120, 160), np.uint8)
    for (y, x) in [(30, 30), (30, 100), (80, 60), (90, 130)]:
        img[y - 6:y + 7, x - 6:x + 7] = 200
    return img


def test_good_features_kats(O):
    img = _blob_image()
    pts = O.good_features(img, None, 100, 0.05, 10, 7)
    assert pts.dtype == np.float32 and pts.shape[1:] == (1, 2)
    xy = pts.reshape(-1, 2)
    assert np.all(xy == np.round(xy))                                       # integer-valued
    assert xy[:, 0].min() >= 1 and xy[:, 1].min() >= 1 and xy[:, 0].max() <= 158 and xy[:, 1].max() <= 118  # never on the border
    d = np.sqrt(((xy[:, None] - xy[None]) ** 2).sum(-1)) + np.eye(len(xy)) * 1e9
    assert d.min() >= 10                                                    # minDistance respected
    corners = np.array([[y + dy, x + dx] for (y, x) in [(30, 30), (30, 100), (80, 60), (90, 130)] for dy in (-6, 6) for dx in (-6, 6)])
    for (x, y) in xy[:16]:
        assert np.abs(corners - [y, x]).sum(1).min() <= 6                  # strongest responses sit on the square corners
    # strongest first; truncation keeps the strongest
    top3 = O.good_features(img, None, 3, 0.05, 10, 7)
    np.testing.assert_array_equal(top3, pts[:3])
    # mask removes corners but never invents new ones
    mask = np.ones_like(img)
    mask[:, :80] = 0
    m = O.good_features(img, mask, 100, 0.05, 10, 7).reshape(-1, 2)
    assert m[:, 0].min() >= 80
    assert O.good_features(np.full((50, 50), 9, np.uint8), None, 10, 0.1, 5, 3) is None   # flat image -> None


def test_good_features_tie_rule_larger_raster_index_first(O):
    # two identical isolated blobs closer than minDistance: exact eig tie, the larger raster index wins
    img = np.zeros((60, 80), np.uint8)
    img[20:25, 20:25] = 255
    img[20:25, 27:32] = 255
    eig = O.min_eigen(img, 5)
    pts = O.good_features(img, None, 0, 0.01, 30, 5).reshape(-1, 2)
    best = np.argwhere(eig == eig.max())
    by, bx = best[np.argmax(best[:, 0] * 80 + best[:, 1])]
    assert len(best) >= 2 and (pts[0] == [bx, by]).all()


def test_pyrdown_vs_scipy(O):
    img = rand_u8((61, 77), seed=5)
    k = np.array([1, 4, 6, 4, 1], np.int64)
    full = ndi.correlate1d(ndi.correlate1d(img.astype(np.int64), k, axis=1, mode="mirror"), k, axis=0, mode="mirror")
    np.testing.assert_array_equal(O.pyrdown_u8(img), ((full[::2, ::2] + 128) >> 8).astype(np.uint8))


def test_lk_kats(O):
    mon, ref = synth.make_pair(200, 240, 0.5, 0.25, noise_sigma=0)
    lm, lr = O.laplacian_u8(O.to_uint8(mon), 7), O.laplacian_u8(O.to_uint8(ref), 7)
    p0 = O.good_features(lr, None, 300, 0.1, 10, 15)
    same, it = O.pyr_lk(lr, lr, p0, return_iters=True)
    np.testing.assert_array_equal(same, p0)                                  # identical images: exactly zero flow
    assert it.max() <= 1
    p1 = O.pyr_lk(lr, lm, p0)
    d = (p1 - p0).reshape(-1, 2)
    assert abs(np.median(d[:, 0]) - 0.5) < 0.05 and abs(np.median(d[:, 1]) - 0.25) < 0.05
    shifted = np.roll(lr, (2, 3), (0, 1))                                    # integer shift (dy=2, dx=3)
    d2 = (O.pyr_lk(lr, shifted, p0) - p0).reshape(-1, 2)
    inner = (p0[:, 0, 0] > 30) & (p0[:, 0, 0] < 210) & (p0[:, 0, 1] > 30) & (p0[:, 0, 1] < 170)
    assert np.abs(np.median(d2[inner], 0) - [3, 2]).max() < 1e-2
    # a point whose window is flat in `prev` keeps its initial guess (not dropped)
    flat = np.zeros((100, 100), np.uint8)
    pt = np.array([[[50.0, 50.0]]], np.float32)
    np.testing.assert_array_equal(O.pyr_lk(flat, rand_u8((100, 100), 3), pt), pt)


def test_config1_pipeline(O, pair512):
    """BASELINE config 1: 512x512 synthetic pair with a known 0.5 px shift, CPU path."""
    mon, ref = pair512
    r = O.klt_tile(mon, ref, O.default_conf())
    assert r["Ninit"] > 1000 and len(r["x0"]) > 0.9 * r["Ninit"]
    assert abs(np.median(r["dx"]) - 0.5) < 0.02 and abs(np.median(r["dy"])) < 0.02
    assert r["score"].min() >= 0 and r["score"].max() <= 1
    assert np.all(np.diff(r["x0"]) >= 0)                                     # sorted by (x0, y0)


def test_mutual_info_golden(O):
    """Both MI scores (next row) against the reference's MutualInfoService / ZNCCService.compute_mi."""
    g = load("mutual_info.npz")
    kp = (g["x0"], g["y0"], g["dx"], g["dy"])
    for ref_key, mon_key, sfx in (("ref", "mon", ""), ("ref_flat", "mon", "_flat"), ("ref_flat", "mon_flat", "_flat2")):
        st, nmi = O.mi_batch(g[ref_key], g[mon_key], *kp)
        for got, exp in ((st, g["studholme" + sfx]), (nmi, g["nmi" + sfx])):
            assert np.array_equal(np.isnan(got), np.isnan(exp))
            assert np.nanmax(np.abs(got - exp)) <= 1e-12
    assert np.nanmin(g["studholme_self"]) == 2.0 and (g["studholme_flat"] == 1.0).any() and (g["nmi_flat"] == 0.0).any()


def test_results_step_oracle_matches_reference_golden():
    """`_handle_klt_results` columns and `_filter_by_dn_values` (api/core.py:650-737, 848-921): the oracle restatement
    against vectors produced by the reference's own methods (tests/golden/make_golden_results.py)."""
    from oracle import oracle as O
    g = load("results.npz")
    ref, mon = g["ref"], g["mon"]
    frames = [{c: g[f"frame{i}_{c}"] for c in ("x0", "y0", "dx", "dy", "score")} for i in range(3)]
    for tag, large in (("scored", False), ("large_shift", True)):
        cols = [O.handle_klt_results_columns(f, ref, mon, 0.4, large) for f in frames]
        names = list(g[f"{tag}_columns"])
        assert names == list(cols[0].keys())
        for name in names:
            got = np.concatenate([c[name] for c in cols])
            want = g[f"{tag}_col_{name}"]
            if name in ("zncc_score", "mutual_info_score", "mi_score"):
                assert np.array_equal(np.isnan(got), np.isnan(want)), name
                np.testing.assert_allclose(got, want, rtol=0, atol=1e-9, equal_nan=True)
            else:
                assert got.dtype == want.dtype and np.array_equal(got, want), name
    x0, y0, idx = g["dn_points_x0"], g["dn_points_y0"], g["dn_points_index"]
    for k in range(int(g["dn_ncases"])):
        nd = g[f"dn_case{k}_nd"]
        keep = O.filter_by_dn_values(x0, y0, ref, mon, list(g[f"dn_case{k}_no_values"]), None if np.isnan(nd[0]) else nd[0],
                                     None if np.isnan(nd[1]) else nd[1])
        assert np.array_equal(idx[keep], g[f"dn_case{k}_kept_index"]), k
        assert np.array_equal(x0[keep], g[f"dn_case{k}_kept_x0"]), k

# Known-answer cases of the reference's tests/test_dn_value_filtering.py (:33-396): 10x10 uint8 images, key points on the
# diagonal, (ref overrides, mon overrides, no_values, expected x0 of the kept key points)
DN_KATS = [
    (dict(fill_ref=1, fill_mon=1, n=5), {}, {}, None, [1, 2, 3, 4, 5]),
    (dict(fill_ref=1, fill_mon=1, n=5), {}, {}, [], [1, 2, 3, 4, 5]),
    (dict(fill_ref=0, fill_mon=50, n=4), {1: 100, 2: 0, 3: 150, 4: 0}, {}, [0], [1, 3]),
    (dict(fill_ref=50, fill_mon=50, n=5), {1: 0, 2: 100, 3: 150, 4: 255, 5: 75}, {}, [0, 100, 255], [3, 5]),
    (dict(fill_ref=100, fill_mon=50, n=4), {}, {1: 0, 2: 50, 3: 0, 4: 50}, [0], [2, 4]),
    (dict(fill_ref=100, fill_mon=50, n=4), {1: 0, 2: 100, 3: 100, 4: 100}, {1: 50, 2: 0, 3: 50, 4: 0}, [0], [3]),
]


def _dn_kat_case(cfg, ref_over, mon_over):
    ref = np.full((10, 10), cfg["fill_ref"], np.uint8)
    mon = np.full((10, 10), cfg["fill_mon"], np.uint8)
    for k, v in ref_over.items():
        ref[k, k] = v
    for k, v in mon_over.items():
        mon[k, k] = v
    xy = np.arange(1, cfg["n"] + 1).astype(np.float32)
    return ref, mon, xy


def test_dn_filter_reference_known_answers():
    from oracle import oracle as O
    for cfg, ro, mo, nv, want in DN_KATS:
        ref, mon, xy = _dn_kat_case(cfg, ro, mo)
        keep = O.filter_by_dn_values(xy, xy, ref, mon, nv)
        assert list(xy[keep].astype(int)) == want


def test_opencv_literal_second_opinion_stays_close(O):
    """oracle/karios_oracle_cvlit.c (OpenCV's float32 evaluation order) vs the exact-integer definition the kernels follow:
    the two may differ by float32 rounding noise only (tools/investigations/oracle_sensitivity.py quantifies it at full size)."""
    from karios_amd import synth
    mon, ref = synth.make_pair(300, 340, 0.5, 0.25, seed=7)
    lap_ref, lap_mon = O.laplacian_u8(O.to_uint8(ref), 7), O.laplacian_u8(O.to_uint8(mon), 7)
    e0 = O.min_eigen(lap_ref, 15)
    strong = e0 > e0.max() * 0.1
    for fma in (False, True):
        e1 = O.min_eigen_cv(lap_ref, 15, fma=fma)
        assert (np.abs(e1 - e0)[strong] / e0[strong]).max() < 5e-6
    p0 = O.select_corners(e0, None, 300, 0.1, 10)
    np.testing.assert_array_equal(p0, O.good_features(lap_ref, None, 300, 0.1, 10, 15))       # select_corners == steps 4-8 of GFTT
    p1 = O.select_corners(O.min_eigen_cv(lap_ref, 15), None, 300, 0.1, 10)
    assert len(set(map(tuple, p0.reshape(-1, 2))) ^ set(map(tuple, p1.reshape(-1, 2)))) <= 4
    a, b = O.pyr_lk(lap_ref, lap_mon, p0), O.pyr_lk_cv(lap_ref, lap_mon, p0)
    assert np.abs(a - b).max() < 5e-3 and np.median(np.abs(a - b)) < 1e-5


# ------------------------------------------------------------------ (c) what REAL OpenCV output confirms
# The reference tree holds one genuine output of cv2 + the reference's glue: the golden CSV of its end-to-end test (inputs
# stripped).  tests/golden/make_golden_csv_facts.py mines it in the build container; the facts below hold without the images.
def _csv_facts():
    import json
    g = load("e2e_csv_facts.npz")
    return json.loads(str(g["facts"])), g


def test_real_opencv_min_distance_is_strict_and_grid_border_rule(O):
    facts, _ = _csv_facts()
    assert facts["corners_are_integers"] and facts["tile_sequence_in_file"] == [0, 1, 2, 3]        # tiles x-outer / y-inner (klt.py:220-232)
    for t in facts["tiles"]:
        assert t["rows"] <= facts["max_corners"] and t["sorted_by_x0_y0"]                           # per-tile maxCorners, (x0, y0) order (klt.py:348)
        # goodFeaturesToTrack rejects a candidate only when dx^2 + dy^2 < minDistance^2: pairs at EXACTLY 10 px exist, none closer
        assert t["nearest_neighbour_min"] == 10.0 and t["nearest_neighbour_below_min_distance"] == 0
        assert t["nearest_neighbour_equal_min_distance"] > 0
        assert t["smallest_distances"][:3] == pytest.approx([10.0, 101 ** 0.5, 104 ** 0.5])
        # corners never sit on the 1-px border of their TILE (the detector runs per tile box), and reach right up to it
        (w, h), (x_lo, x_hi), (y_lo, y_hi) = t["size"], t["local_x_range"], t["local_y_range"]
        assert 1 <= x_lo and x_hi <= w - 2 and 1 <= y_lo and y_hi <= h - 2
    full = facts["tiles"][0]
    assert full["local_x_range"] == [1.0, 5998.0] and full["local_y_range"] == [1.0, 5998.0]
    # the oracle's selection has exactly that rule: a peak 10 px from a stronger one survives, one at sqrt(98) px does not
    eig = np.zeros((64, 64), np.float32)
    eig[20, 20], eig[28, 26], eig[27, 27 + 20] = 3.0, 2.0, 1.0                 # (26,28) is exactly 10 px from (20,20); (47,27) is far
    eig[27, 27] = 1.5                                                          # sqrt(98) px from (20,20), and < 10 px from (26,28)
    got = O.select_corners(eig, None, 0, 0.01, 10).reshape(-1, 2).tolist()
    assert got == [[20.0, 20.0], [26.0, 28.0], [47.0, 27.0]]


def test_real_opencv_tracks_are_float32_in_tile_coordinates():
    """dx = x1 - x0 and d = max|p0 - p0r| are float32 operations on TILE-LOCAL coordinates (the tile offset is added afterwards,
    klt.py:341-342): x0_local + dx is exactly representable in float32 for every row, x0_global + dx is not; every score is
    1 - d / float32(0.1) for a d on the float32 grid of the local coordinate - reproduced through karios_amd.frames."""
    from karios_amd import frames
    facts, g = _csv_facts()
    x, y, dx, dy, score, tile = (g[k] for k in ("x0", "y0", "dx", "dy", "score", "tile"))
    T = facts["tile_size"]
    xl, yl = x - (tile // 2) * np.float32(T), y - (tile % 2) * np.float32(T)

    def representable(a64):
        return a64.astype(np.float32).astype(np.float64) == a64
    assert representable(xl.astype(np.float64) + dx).all() and representable(yl.astype(np.float64) + dy).all()
    off = tile >= 2
    assert representable(x[off].astype(np.float64) + dx[off]).mean() < 0.5       # not so in image coordinates
    # score lattice: search the float32 grid of the local coordinates for the forward-backward distance behind each score
    limit = frames.FB_LIMIT
    found = np.zeros(len(score), bool)
    p0 = np.stack([xl, yl], 1)
    p0r = p0.copy()
    for axis, coord in enumerate((xl, yl)):
        for fine in (1.0, 0.5):                                                # just below a power of two the grid is twice as fine
            u = np.spacing(coord.astype(np.float32)).astype(np.float64) * fine
            k0 = np.round((1.0 - score.astype(np.float64)) * 0.1 / u)
            for dk in (-2, -1, 0, 1, 2):
                back = (coord.astype(np.float64) + (k0 + dk) * u * (1 if fine == 1.0 else -1)).astype(np.float32)   # a return point on that grid
                d = np.abs(coord - back)
                hit = ((np.float32(1) - d / limit) == score) & (d < limit) & ~found
                p0r[hit, axis], found = back[hit], found | hit
    assert found.all()
    # ... and the product's own track -> frame arithmetic returns the CSV's columns from such tracks
    p1 = np.stack([xl + dx, yl + dy], 1).astype(np.float32)
    cols, n = frames.track_columns(p0.reshape(-1, 1, 2), p1.reshape(-1, 1, 2), p0r.reshape(-1, 1, 2))
    assert n == len(score) and len(cols["score"]) == len(score)                 # all pass the FB test (they are in the CSV)
    np.testing.assert_array_equal(cols["score"], score)
    np.testing.assert_array_equal(cols["dx"], dx)
    np.testing.assert_array_equal(cols["dy"], dy)
    for t in range(4):
        m = tile == t
        f = frames.assemble({k: v[m] for k, v in cols.items()}, (t // 2) * T, (t % 2) * T)
        assert f["x0"].dtype == np.float32 and np.array_equal(np.sort(f["x0"].to_numpy()), np.sort(x[m]))


def test_real_opencv_result_columns_and_zncc_bounds_rule():
    """`radial error` recomputes bit-exactly and `angle` to numpy's arctan2 accuracy from the float32 dx, dy (core.py:872-873);
    the NaN pattern of `zncc_score` is exactly {score < 0.4} + the chip bounds rule - with ONE documented difference: the run
    that wrote the CSV predates the `>=` of zncc_service.py:212-215 and scored the two rows at y0 == size - 28."""
    import pandas as pd
    from karios_amd import frames
    from karios_amd.matcher.zncc_service import CHIP_SIZE, _chip_centres
    facts, g = _csv_facts()
    frame = pd.DataFrame({k: g[k] for k in ("x0", "y0", "dx", "dy", "score")})
    out = frames.radial_angle_columns(frame.copy())
    np.testing.assert_array_equal(out["radial error"].to_numpy(), g["radial_error"])
    assert out["angle"].dtype == np.float32
    assert (np.abs(out["angle"].to_numpy() - g["angle"]) <= 4 * np.spacing(np.abs(g["angle"]))).all()
    S = facts["image_size"]
    *_, inside = _chip_centres(frame, (CHIP_SIZE - 1) // 2, (S, S), (S, S))
    expect_nan = (g["score"] < np.float32(0.4)) | ~inside
    csv_nan = np.isnan(g["zncc_score"])
    differ = np.flatnonzero(expect_nan != csv_nan)
    assert len(differ) == 2 and (g["y0"][differ] == S - 28).all() and not csv_nan[differ].any()
    assert not (csv_nan & ~expect_nan).any()                                     # no NaN the rule does not explain (e.g. flat windows)


def _oscillation_quads():
    """(ddx, pdx, ddy, pdy) rows whose float32 sums sit exactly on, one ulp below and one ulp above float32(0.01), and OpenCV's verdict:
    `std::abs(delta.x + prevDelta.x) < 0.01` compares a float32 with the DOUBLE literal - float32(0.01) = 0.00999999977... passes."""
    f = np.float32
    edge = f(0.01)
    below, above = np.nextafter(edge, f(0)), np.nextafter(edge, f(1))
    rows, want = [], []
    rng = np.random.default_rng(5)
    for target, verdict in ((edge, True), (below, True), (above, False), (-edge, True), (-above, False)):
        found = 0
        while found < 8:
            p = f(rng.uniform(-0.02, 0.02))
            d = f(target - p)
            if f(d + p) != target:
                continue
            rows.append((d, p, f(0.001), f(-0.0005))); want.append(verdict)       # x decides
            rows.append((f(0.001), f(-0.0005), d, p)); want.append(verdict)       # y decides
            found += 1
    return np.array(rows, np.float32), np.array(want)


def test_lk_oscillation_literal_is_double(O):
    """OpenCV's oscillation stop uses the double literal 0.01; a float32 sum of exactly float32(0.01) stops the
    iteration (it would not with `< 0.01f`).  Both oracles' predicates (the one the tracker loops call) are checked."""
    q, want = _oscillation_quads()
    assert float(np.float32(0.01)) < 0.01 and float(np.nextafter(np.float32(0.01), np.float32(1))) > 0.01
    for lit in (False, True):
        got = np.array([O.lk_oscillates(*row, literal_oracle=lit) for row in q])
        np.testing.assert_array_equal(got, want)
