#!/usr/bin/env python3
"""How far can the two defensible roundings of the OpenCV-defined arithmetic move the matcher's result?

The kernels (and oracle/karios_oracle.c) DEFINE goodFeaturesToTrack's structure tensor and PyrLK's normal matrix / mismatch
vector through exact integer sums; OpenCV (absent: "parity unpinned") rounds to float32 earlier.
`oracle/karios_oracle_cvlit.c` restates OpenCV's own float32 evaluation order.  This script runs both on the same synthetic
pair (default: BASELINE config 2, 10980 x 10980, SURVEY 8d generator) and reports

  * min-eigenvalue map: relative difference over the candidate range;
  * goodFeaturesToTrack: corners in / out of the selected set, rank of the first divergence, and for every flipped decision
    the relative eigenvalue gap of the two corners that competed (a flip needs the gap to be within the rounding noise);
  * calcOpticalFlowPyrLK forward / backward on the common corners: max |d dx|, |d dy|, |d score|, rows whose forward-backward
    verdict differs.

CPU only (the oracle), a few minutes at full size.  Writes profiles/r02_oracle_sensitivity.json, which bench.py attaches to its
JSON line as `oracle_sensitivity`.

    python tools/investigations/oracle_sensitivity.py [--size 10980] [--out profiles/r02_oracle_sensitivity.json]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def corner_report(O, e_ref, p_ref, p_alt, min_distance, max_corners):
    a = p_ref.reshape(-1, 2).astype(np.int64)
    b = p_alt.reshape(-1, 2).astype(np.int64)
    sa, sb = set(map(tuple, a)), set(map(tuple, b))
    only_a = np.array(sorted(sa - sb)).reshape(-1, 2)
    only_b = np.array(sorted(sb - sa)).reshape(-1, 2)
    n = min(len(a), len(b))
    differ = np.flatnonzero((a[:n] != b[:n]).any(axis=1))
    rep = {"corners_exact": len(a), "corners_literal": len(b), "in_common": len(sa & sb), "only_exact": len(only_a), "only_literal": len(only_b),
           "rank_of_first_divergence": (int(differ[0]) if len(differ) else None),
           "order_identical": bool(len(a) == len(b) and not len(differ))}
    if len(only_a) or len(only_b):
        val = lambda pts: e_ref[pts[:, 1], pts[:, 0]].astype(np.float64)
        weakest = float(val(a).min())
        gaps, causes = [], {"conflict_within_minDistance": 0, "cut_at_maxCorners_or_threshold": 0}
        from scipy.spatial import cKDTree
        tree = cKDTree(only_b) if len(only_b) else None
        for p in only_a:
            partner = None
            if tree is not None:
                d, j = tree.query(p, k=1)
                if d < min_distance:
                    partner = only_b[j]
            if partner is not None:
                va, vb = val(p[None])[0], val(partner[None])[0]
                gaps.append(abs(va - vb) / max(abs(va), abs(vb)))
                causes["conflict_within_minDistance"] += 1
            else:
                va = val(p[None])[0]
                gaps.append(abs(va - weakest) / max(abs(va), 1e-30))
                causes["cut_at_maxCorners_or_threshold"] += 1
        rep["flipped_decisions"] = causes
        rep["relative_eig_gap_of_flips"] = {"min": float(np.min(gaps)), "median": float(np.median(gaps)), "max": float(np.max(gaps))}
    return rep


def lk_report(O, lap_ref, lap_mon, p0, win):
    out = {}
    fwd = {"exact": O.pyr_lk(lap_ref, lap_mon, p0, win), "literal": O.pyr_lk_cv(lap_ref, lap_mon, p0, win)}
    bwd = {"exact": O.pyr_lk(lap_mon, lap_ref, fwd["exact"], win), "literal": O.pyr_lk_cv(lap_mon, lap_ref, fwd["literal"], win)}
    d = {k: np.abs(p0 - bwd[k]).reshape(-1, 2).max(-1) for k in fwd}
    keep = {k: d[k] < np.float32(0.1) for k in fwd}
    both = keep["exact"] & keep["literal"]
    disp = {k: (fwd[k] - p0).reshape(-1, 2) for k in fwd}
    dd = np.abs(disp["exact"] - disp["literal"])
    score = {k: 1 - d[k] / np.float32(0.1) for k in fwd}
    out["points"] = int(len(p0))
    out["kept_exact"], out["kept_literal"] = int(keep["exact"].sum()), int(keep["literal"].sum())
    out["forward_backward_verdict_differs"] = int((keep["exact"] != keep["literal"]).sum())
    out["max_abs_ddx_ddy_px_kept_in_both"] = [float(dd[both, 0].max()), float(dd[both, 1].max())] if both.any() else None
    out["p999_abs_d_px_kept_in_both"] = float(np.quantile(dd[both].max(axis=1), 0.999)) if both.any() else None
    out["max_abs_dscore_kept_in_both"] = float(np.abs(score["exact"] - score["literal"])[both].max()) if both.any() else None
    out["max_abs_d_px_all_points"] = float(dd.max())
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=10980)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r02_oracle_sensitivity.json"))
    a = ap.parse_args()
    from karios_amd import synth
    from oracle import oracle as O
    O.build()
    t0 = time.time()
    S = a.size
    mon, ref = synth.make_pair(S, S, 0.5, 0.25)                      # BASELINE config 2 generator
    conf = O.default_conf()
    mask, _ = O.auto_mask(mon, ref)
    lap_ref = O.laplacian_u8(O.to_uint8(ref), 7)
    lap_mon = O.laplacian_u8(O.to_uint8(mon), 7)
    print(f"inputs ready after {time.time() - t0:.0f} s", flush=True)
    e_exact = O.min_eigen(lap_ref, conf.blocksize)
    report = {"workload": f"synthetic S2 pair {S}x{S} (SURVEY 8d generator, shift 0.5/0.25), Laplacian k=7, maxCorners {conf.maxCorners}, "
                          f"qualityLevel {conf.qualityLevel}, minDistance {conf.minDistance}, blockSize {conf.blocksize}, winSize {conf.matching_winsize}",
              "what": "exact-integer definition (kernels + oracle) vs OpenCV-literal float32 evaluation (oracle/karios_oracle_cvlit.c)"}
    p_exact = O.select_corners(e_exact, mask, conf.maxCorners, conf.qualityLevel, conf.minDistance)
    strong = e_exact > e_exact.max() * conf.qualityLevel
    for label, fma in (("literal", False), ("literal_fma", True)):
        e_alt = O.min_eigen_cv(lap_ref, conf.blocksize, fma=fma)
        rel = np.abs(e_alt.astype(np.float64) - e_exact) / np.abs(e_exact.astype(np.float64)).clip(1e-30)
        p_alt = O.select_corners(e_alt, mask, conf.maxCorners, conf.qualityLevel, conf.minDistance)
        rep = {"eig_relative_difference_above_threshold": {"max": float(rel[strong].max()), "median": float(np.median(rel[strong]))}}
        rep.update(corner_report(O, e_exact, p_exact, p_alt, conf.minDistance, conf.maxCorners))
        report[f"good_features_{label}"] = rep
        print(label, json.dumps(rep), flush=True)
        del e_alt
    report["pyr_lk"] = lk_report(O, lap_ref, lap_mon, p_exact, conf.matching_winsize)
    print("pyr_lk", json.dumps(report["pyr_lk"]), flush=True)
    lk = report["pyr_lk"]
    report["north_star_1e-3_px_holds_between_the_two"] = bool(lk["max_abs_ddx_ddy_px_kept_in_both"] is not None and
                                                             max(lk["max_abs_ddx_ddy_px_kept_in_both"]) <= 1e-3)
    report["seconds"] = round(time.time() - t0, 1)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(report, open(a.out, "w"), indent=1)
    print("written", a.out)


if __name__ == "__main__":
    main()
