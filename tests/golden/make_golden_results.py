#!/usr/bin/env python3
"""Golden vectors for the post-matching step, produced by the REFERENCE's own methods
`KariosAPI._handle_klt_results` (karios/api/core.py:848-921) and `KariosAPI._filter_by_dn_values` (core.py:650-737),
imported from /root/reference (build container only) and called unbound on a minimal stand-in for `self`.

Run here, never on the GPU box:   python tests/golden/make_golden_results.py      -> tests/golden/results.npz
Only DATA (inputs + expected outputs, the CSV as text) is written; no reference source is copied.
"""
import os
import sys
import tempfile
import types

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import Img, import_reference  # noqa: E402


def main():
    from karios_amd import synth
    _, rzncc, _, rmi = import_reference()
    import karios.api.core as rcore

    rng = np.random.default_rng(20261003)
    mon, ref = synth.make_pair(230, 270, 0.4, -0.3, seed=11)
    mon, ref = mon.copy(), ref.copy()
    # DN values the filter looks for, under some key points
    ref[40:44, 50:60] = 0
    mon[100:104, 120:130] = 0
    ref[150, 20:40] = 1234
    mon[10:14, 200:210] = 777
    out = {"ref": ref, "mon": mon}

    def kp_frame(n, seed, lo=0):
        r = np.random.default_rng(seed)
        x0 = r.integers(0, 270, n).astype(np.float32)
        y0 = r.integers(0, 230, n).astype(np.float32)
        x0[:4] = [0, 269, 30, 27]          # chips leaving the image -> NaN scores
        y0[:4] = [0, 229, 28, 100]
        dx = (0.4 + 0.3 * r.standard_normal(n)).astype(np.float32)
        dy = (-0.3 + 0.3 * r.standard_normal(n)).astype(np.float32)
        score = r.random(n).astype(np.float32)
        score[4:8] = [0.4, 0.39999998, 1.0, 0.0]
        order = np.lexsort((y0, x0))
        idx = r.permutation(n)[order] + lo      # index labels as pandas leaves them after sort_values
        return pd.DataFrame({"x0": x0[order], "y0": y0[order], "dx": dx[order], "dy": dy[order], "score": score[order]}, index=idx)

    frames = [kp_frame(60, 1), kp_frame(45, 2), kp_frame(30, 3)]
    for i, f in enumerate(frames):
        for c in f.columns:
            out[f"frame{i}_{c}"] = f[c].to_numpy()
        out[f"frame{i}_index"] = f.index.to_numpy()

    def fake_self(large):
        s = types.SimpleNamespace()
        s._large_shift_applied = large
        s._processing_configuration = types.SimpleNamespace(accuracy_analysis_configuration=types.SimpleNamespace(confidence_threshold=0.4))
        s._zncc_service = rzncc.ZNCCService()
        s._mutual_info_service = rmi.MutualInfoService()
        return s

    for tag, large in (("scored", False), ("large_shift", True)):
        with tempfile.TemporaryDirectory() as td:
            from pathlib import Path
            csv = Path(td) / "kp.csv"
            res = rcore.KariosAPI._handle_klt_results(fake_self(large), iter([f.copy() for f in frames]), csv, Img(mon), Img(ref))
            out[f"{tag}_csv"] = np.frombuffer(csv.read_bytes(), np.uint8)
            out[f"{tag}_columns"] = np.array(list(res.columns))
            out[f"{tag}_index"] = res.index.to_numpy()
            for c in res.columns:
                out[f"{tag}_col_{c}"] = res[c].to_numpy()

    # ---- _filter_by_dn_values
    pts = pd.concat(frames)
    cases = [dict(no_values=None, ref_nd=None, mon_nd=None), dict(no_values=[0], ref_nd=None, mon_nd=None),
             dict(no_values=[0, 1234], ref_nd=None, mon_nd=777.0), dict(no_values=[], ref_nd=1234.0, mon_nd=None),
             dict(no_values=[777, 5], ref_nd=0.0, mon_nd=0.0), dict(no_values=[70000, -3], ref_nd=None, mon_nd=None)]
    # make sure some key points sit on the special pixels
    pts = pts.copy()
    special = [(55, 41), (125, 101), (30, 150), (205, 12), (56, 42)]
    for j, (x, y) in enumerate(special):
        pts.iloc[j, pts.columns.get_loc("x0")] = np.float32(x)
        pts.iloc[j, pts.columns.get_loc("y0")] = np.float32(y)
    out["dn_points_x0"], out["dn_points_y0"] = pts["x0"].to_numpy(), pts["y0"].to_numpy()
    out["dn_points_index"] = pts.index.to_numpy()
    for k, cs in enumerate(cases):
        got = rcore.KariosAPI._filter_by_dn_values(types.SimpleNamespace(), pts, Img(mon, cs["mon_nd"]), Img(ref, cs["ref_nd"]), cs["no_values"])
        out[f"dn_case{k}_no_values"] = np.array(cs["no_values"] if cs["no_values"] else [], np.float64)
        out[f"dn_case{k}_nd"] = np.array([np.nan if cs["ref_nd"] is None else cs["ref_nd"], np.nan if cs["mon_nd"] is None else cs["mon_nd"]])
        out[f"dn_case{k}_kept_index"] = got.index.to_numpy()
        out[f"dn_case{k}_kept_x0"] = got["x0"].to_numpy()
    out["dn_ncases"] = np.array(len(cases))
    np.savez_compressed(os.path.join(HERE, "results.npz"), **out)
    print("results.npz written:", {k: v.shape for k, v in out.items() if k.endswith("_csv")})


if __name__ == "__main__":
    main()
