"""GPU vs the committed golden vectors (outputs of the reference's own Python functions) and the
drop-in `karios_amd.matcher` classes against the reference's KLT.match / ZNCCService results."""
import os

import numpy as np
import pandas as pd
import pytest

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return np.load(os.path.join(G, name))


def test_to_uint8_golden(ops):
    g = load("to_uint8.npz")
    for k in [k[3:] for k in g.files if k.startswith("in_")]:
        np.testing.assert_array_equal(ops.to_uint8(g["in_" + k]), g["out_" + k], err_msg=k)


def test_shift_image_golden(ops):
    from karios_amd.core import shift_image
    g = load("shift_image.npz")
    for i, (yo, xo) in enumerate(g["offsets"]):
        out = shift_image(g["img"], y_off=yo, x_off=xo)
        assert out.dtype == g["img"].dtype
        np.testing.assert_array_equal(out, g[f"out_{i}"])


def test_zncc_service_golden(ops):
    from karios_amd.core import NumpyRasterImage
    from karios_amd.matcher import ZNCCService
    g = load("zncc.npz")
    df = pd.DataFrame({c: g[c] for c in ("x0", "y0", "dx", "dy")})
    df.index = df.index * 3 + 7                       # the Series must keep the frame's index
    for ref_key, exp_key in (("ref", "zncc"), ("ref_flat", "zncc_flat")):
        s = ZNCCService().compute_zncc(df, NumpyRasterImage(g["mon"]), NumpyRasterImage(g[ref_key]))
        assert s.index.equals(df.index) and s.dtype == np.float64
        exp = g[exp_key]
        assert np.array_equal(np.isnan(s.to_numpy()), np.isnan(exp))
        assert np.nanmax(np.abs(s.to_numpy() - exp)) <= 1e-9
    empty = ZNCCService().compute_zncc(df.iloc[:0], NumpyRasterImage(g["mon"]), NumpyRasterImage(g["ref"]))
    assert len(empty) == 0


CONFS = {
    "tiles": dict(tile_size=200, maxCorners=600, laplacian_kernel_size=7),
    "xstart": dict(tile_size=130, xStart=130, maxCorners=300, laplacian_kernel_size=5),
    "mixed_inv": dict(tile_size=20000, maxCorners=800, laplacian_kernel_size={"mon": 5, "ref": 9},
                      laplacian_invert_polarity=True, outliers_filtering=True),
    "usermask": dict(tile_size=20000, maxCorners=500, laplacian_kernel_size=3),
    "auto_ksize": dict(tile_size=20000, maxCorners=250, laplacian_kernel_size="auto"),
    "auto_polarity": dict(tile_size=120, maxCorners=200, laplacian_kernel_size=5, laplacian_invert_polarity="auto"),
}


@pytest.mark.parametrize("case", list(CONFS))
@pytest.mark.parametrize("gen_laplacian", [False, True])
def test_klt_match_drop_in_golden(ops, case, gen_laplacian, tmp_path):
    """karios_amd.matcher.KLT.match == reference KLT.match (with the oracle behind cv2): same frames in
    the same order, key-point coordinates bit-exact, displacements within 1e-3 px, score within 1e-2.
    gen_laplacian=True exercises the unfused (operator by operator) path."""
    from karios_amd.core import KLTConfiguration, NumpyRasterImage
    from karios_amd.matcher import KLT
    g = load(f"klt_match_{case}.npz")
    nd = [None if np.isnan(v) else v for v in g["nodata"]]
    mask = NumpyRasterImage(g["mask"]) if "mask" in g.files else None
    klt = KLT(KLTConfiguration(**CONFS[case]), gen_laplacian=gen_laplacian, out_dir=str(tmp_path))
    frames = list(klt.match(NumpyRasterImage(g["mon"], nd[0]), NumpyRasterImage(g["ref"], nd[1]), mask))
    assert len(frames) == int(g["n_frames"])
    for i, f in enumerate(frames):
        assert list(f.columns) == ["x0", "y0", "dx", "dy", "score"] and all(f[c].dtype == np.float32 for c in f.columns)
        np.testing.assert_array_equal(f["x0"].to_numpy(), g[f"f{i}_x0"])
        np.testing.assert_array_equal(f["y0"].to_numpy(), g[f"f{i}_y0"])
        assert np.abs(f["dx"].to_numpy() - g[f"f{i}_dx"]).max() <= 1e-3
        assert np.abs(f["dy"].to_numpy() - g[f"f{i}_dy"]).max() <= 1e-3
        assert np.abs(f["score"].to_numpy() - g[f"f{i}_score"]).max() <= 1e-2
    if "auto_ksize" in g.files and case == "auto_ksize":
        assert klt.auto_selected_ksize == tuple(int(v) for v in g["auto_ksize"])
    if case == "auto_polarity":
        assert klt.auto_selected_polarity == str(g["auto_polarity"])
    if gen_laplacian:
        assert len(os.listdir(tmp_path)) == 2 * len(KLT(KLTConfiguration(**CONFS[case])).tile_boxes(g["mon"].shape[1], g["mon"].shape[0]))


def test_phase_correlation_golden(ops):
    from karios_amd.core import NumpyRasterImage
    from karios_amd.matcher import LargeOffsetMatcher
    g = load("phase_corr.npz")
    ref = g["ref"]
    for s, exp in zip(g["shifts"], g["skimage_shift"]):
        mon = np.roll(ref, tuple(s), (0, 1))
        got = LargeOffsetMatcher(NumpyRasterImage(ref), NumpyRasterImage(mon)).match()
        np.testing.assert_array_equal(got, exp)


def test_mutual_info_services_golden(ops):
    """MutualInfoService.compute_mutual_info and ZNCCService.compute_mi on the GPU vs the reference's outputs:
    same NaN pattern (bounds, zero entropy), values within 1e-9 (fp64; log from the device libm)."""
    from karios_amd.core import NumpyRasterImage
    from karios_amd.matcher import MutualInfoService, ZNCCService
    g = load("mutual_info.npz")
    df = pd.DataFrame({c: g[c] for c in ("x0", "y0", "dx", "dy")})
    df.index = df.index + 100
    for ref_key, mon_key, sfx in (("ref", "mon", ""), ("ref_flat", "mon", "_flat"), ("ref_flat", "mon_flat", "_flat2")):
        mon, ref = NumpyRasterImage(g[mon_key]), NumpyRasterImage(g[ref_key])
        st = MutualInfoService().compute_mutual_info(df, mon, ref)
        nmi = ZNCCService().compute_mi(df, mon, ref)
        for got, exp in ((st, g["studholme" + sfx]), (nmi, g["nmi" + sfx])):
            assert got.index.equals(df.index)
            v = got.to_numpy()
            assert np.array_equal(np.isnan(v), np.isnan(exp)), sfx
            assert np.nanmax(np.abs(v - exp)) <= 1e-9, sfx
    z0 = pd.DataFrame({c: g[c] for c in ("x0", "y0")}).assign(dx=np.float32(0), dy=np.float32(0)).astype(np.float32)
    same = MutualInfoService().compute_mutual_info(z0, NumpyRasterImage(g["ref"]), NumpyRasterImage(g["ref"])).to_numpy()
    assert np.array_equal(np.isnan(same), np.isnan(g["studholme_self"])) and np.nanmax(np.abs(same - 2.0)) <= 1e-12


@pytest.mark.parametrize("dtype", [np.uint8, np.int16, np.float32])
def test_mutual_info_other_dtypes(ops, dtype):
    from oracle import oracle as O
    g = load("mutual_info.npz")
    ref, mon = (g["ref"] // 48).astype(dtype), (g["mon"] // 48).astype(dtype)
    if dtype == np.float32:
        ref, mon = ref + np.float32(0.25), mon * np.float32(1.5)
    kp = (g["x0"], g["y0"], g["dx"], g["dy"])
    st, nmi = ops.mi_batch(ref, mon, *kp)
    est, enmi = O.mi_batch(ref, mon, *kp)
    for got, exp in ((st, est), (nmi, enmi)):
        assert np.array_equal(np.isnan(got), np.isnan(exp)) and np.nanmax(np.abs(got - exp)) <= 1e-9


def test_results_step_matches_reference_golden(tmp_path):
    """SURVEY 8(f)-2 on the device: `karios_amd.results.handle_klt_results` / `filter_by_dn_values` against the vectors the
    reference's `_handle_klt_results` / `_filter_by_dn_values` produced (CSV structure, column order and dtypes, float32
    columns bit-exact, scores to 1e-9)."""
    import io
    import pandas as pd
    from karios_amd.resident import ResidentPair
    from karios_amd.results import filter_by_dn_values, handle_klt_results
    g = load("results.npz")
    ref, mon = g["ref"], g["mon"]
    pair = ResidentPair.upload(mon, ref)

    def frames():
        return [pd.DataFrame({c: g[f"frame{i}_{c}"] for c in ("x0", "y0", "dx", "dy", "score")}, index=g[f"frame{i}_index"]) for i in range(3)]

    for tag, large in (("scored", False), ("large_shift", True)):
        csv = tmp_path / f"{tag}.csv"
        res = handle_klt_results(iter(frames()), csv, pair, 0.4, large_shift_applied=large)
        names = list(g[f"{tag}_columns"])
        assert list(res.columns) == names
        assert np.array_equal(res.index.to_numpy(), g[f"{tag}_index"])
        for name in names:
            got, want = res[name].to_numpy(), g[f"{tag}_col_{name}"]
            if name in ("zncc_score", "mutual_info_score", "mi_score"):
                assert got.dtype == np.float64 and np.array_equal(np.isnan(got), np.isnan(want)), name
                np.testing.assert_allclose(got, want, rtol=0, atol=1e-9, equal_nan=True)
            else:
                assert got.dtype == want.dtype and np.array_equal(got, want), name
        want_csv = pd.read_csv(io.BytesIO(g[f"{tag}_csv"].tobytes()), sep=";")
        got_csv = pd.read_csv(csv, sep=";")
        assert list(got_csv.columns) == list(want_csv.columns) and len(got_csv) == len(want_csv)
        text_got, text_want = csv.read_text().splitlines(), g[f"{tag}_csv"].tobytes().decode().splitlines()
        assert text_got[0] == text_want[0] and len(text_got) == len(text_want)      # header once, one line per key point
        for a, b in zip(text_got[1:], text_want[1:]):
            assert a.split(";")[:7] == b.split(";")[:7]                                # float32 columns: identical text
            assert [v == "" for v in a.split(";")] == [v == "" for v in b.split(";")]  # NaN cells are empty in both
        pd.testing.assert_frame_equal(got_csv, want_csv, rtol=0, atol=1e-9)

    pts = pd.DataFrame({"x0": g["dn_points_x0"], "y0": g["dn_points_y0"]}, index=g["dn_points_index"])
    for k in range(int(g["dn_ncases"])):
        nd = g[f"dn_case{k}_nd"]
        pair.no_data_ref = None if np.isnan(nd[0]) else float(nd[0])
        pair.no_data_mon = None if np.isnan(nd[1]) else float(nd[1])
        got = filter_by_dn_values(pts, pair, list(g[f"dn_case{k}_no_values"]))
        assert np.array_equal(got.index.to_numpy(), g[f"dn_case{k}_kept_index"]), k
        assert np.array_equal(got["x0"].to_numpy(), g[f"dn_case{k}_kept_x0"]), k
    bad = pd.DataFrame({"x0": np.float32([5, 999]), "y0": np.float32([5, 5])})
    pair.no_data_ref = 0.0
    with pytest.raises(IndexError):
        filter_by_dn_values(bad, pair, [1])


def test_dn_filter_reference_known_answers():
    """The six cases of the reference's tests/test_dn_value_filtering.py on the device."""
    from test_oracle_golden import DN_KATS, _dn_kat_case
    from karios_amd.resident import ResidentPair
    from karios_amd.results import filter_by_dn_values
    for cfg, ro, mo, nv, want in DN_KATS:
        ref, mon, xy = _dn_kat_case(cfg, ro, mo)
        pts = pd.DataFrame({"x0": xy, "y0": xy, "dx": xy * 0.1, "dy": xy * 0.1, "score": 1 - xy * 0.1})
        got = filter_by_dn_values(pts, ResidentPair.upload(mon, ref), nv)
        assert list(got["x0"].astype(int)) == want and list(got["y0"].astype(int)) == want
        if not nv:
            assert got is pts                      # nothing requested: the frame comes back untouched (core.py:673-675)
