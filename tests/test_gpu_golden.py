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
