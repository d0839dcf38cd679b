"""The CPU legs of the bench: the oracle timed as `cpu_baseline` (kind 'port') and the parity gates on the measured pairs.
Only this module, `benchkit.sensitivity` and config 3's crop gate touch `oracle/` - always outside timed regions."""
from __future__ import annotations

import json
import os
import statistics
import time

import numpy as np

def parity_gate(frame, res, zncc_oracle):
    """SURVEY 8(d) parity gates on the pair that was measured: the GPU frame of the timed loop against the oracle's result for the
    same full-size pair - key points identical and in the same order, displacements within 1e-3 px, score within 1e-2, ZNCC within 1e-9."""
    if frame is None or res is None:
        return {"checked": False}
    gx, gy = frame["x0"].to_numpy(), frame["y0"].to_numpy()
    same = len(gx) == len(res["x0"]) and bool(np.array_equal(gx, res["x0"]) and np.array_equal(gy, res["y0"]))
    out = {"checked": True, "rows_gpu": int(len(gx)), "rows_oracle": int(len(res["x0"])), "keypoints_identical_and_in_order": same}
    if same:
        out["max_abs_ddx_px"] = float(np.abs(frame["dx"].to_numpy() - res["dx"]).max())
        out["max_abs_ddy_px"] = float(np.abs(frame["dy"].to_numpy() - res["dy"]).max())
        out["max_abs_dscore"] = float(np.abs(frame["score"].to_numpy() - res["score"]).max())
        if zncc_oracle is not None and "zncc_score" in frame.columns:
            keep = res["score"] >= 0.4
            z = frame["zncc_score"].to_numpy()[keep]
            out["zncc_nan_pattern_identical"] = bool(np.array_equal(np.isnan(z), np.isnan(zncc_oracle)))
            both = ~np.isnan(z) & ~np.isnan(zncc_oracle)
            out["max_abs_dzncc"] = float(np.abs(z[both] - zncc_oracle[both]).max()) if both.any() else 0.0
        out["passed"] = bool(out["max_abs_ddx_px"] <= 1e-3 and out["max_abs_ddy_px"] <= 1e-3 and out["max_abs_dscore"] <= 1e-2
                             and out.get("zncc_nan_pattern_identical", True) and out.get("max_abs_dzncc", 0.0) <= 1e-9)
    else:
        out["passed"] = False
    return out


def full_scoring_gate(O, mon, ref, frame):
    """core.py:894-907 on the measured pair: the device's `mutual_info_score` / `mi_score` / `zncc_score` columns of EVERY row against
    the oracle evaluated on the frame's own key points (their identity with the oracle's is the headline gate): NaN exactly where
    score < 0.4 or the chip leaves the image, <= 1e-9 elsewhere."""
    if frame is None or "mutual_info_score" not in frame.columns:
        return {"checked": False}
    x0, y0, dx, dy, sc = (frame[c].to_numpy() for c in ("x0", "y0", "dx", "dy", "score"))
    keep = sc >= np.float32(0.4)
    out = {"checked": True, "rows": int(len(frame)), "rows_scored": int(keep.sum())}
    want = {"zncc_score": np.full(len(frame), np.nan), "mutual_info_score": np.full(len(frame), np.nan), "mi_score": np.full(len(frame), np.nan)}
    if keep.any():
        want["zncc_score"][keep] = O.zncc_batch(ref, mon, x0[keep], y0[keep], dx[keep], dy[keep])
        st, nmi = O.mi_batch(ref, mon, x0[keep], y0[keep], dx[keep], dy[keep])
        want["mutual_info_score"][keep], want["mi_score"][keep] = st, nmi
    ok = True
    for col, w in want.items():
        g = frame[col].to_numpy()
        same_nan = bool(np.array_equal(np.isnan(g), np.isnan(w)))
        both = ~np.isnan(g) & ~np.isnan(w)
        err = float(np.abs(g[both] - w[both]).max()) if both.any() else 0.0
        out[col] = {"nan_pattern_identical": same_nan, "max_abs_diff": err, "finite_rows": int(both.sum())}
        ok = ok and same_nan and err <= 1e-9
    out["tolerance"] = 1e-9
    out["passed"] = bool(ok)
    return out


def cpu_baseline(get_pair, n_pairs, conf_kw, runs, gpu_frames=None, scored_frame=None):
    """Oracle (kind 'port') on the SAME full pairs the GPU loop ran, all usable cores: one timed pass per DISTINCT pair (`runs` passes,
    pair b = pass b % n_pairs; the median pass is the figure), each pass gating that pair's GPU frame of the timed loop (`gpu_frames`:
    pair index -> DataFrame); plus a 1-thread figure on the top tenth of pair 0 (maxCorners scaled to the same corner density).
    `get_pair(b)` -> (mon, ref) host arrays of pair b (482 MB each pair: fetched one at a time)."""
    from oracle import oracle as O
    conf = O.default_conf(**conf_kw)
    cores = min(O.usable_cpus(), int(os.environ.get("KARIOS_ORACLE_THREADS", "1024")))
    O.set_threads(cores)

    def one_pass(m, r, c):
        t0 = time.perf_counter()
        res = O.klt_tile(m, r, c)
        n = 0
        if res is not None:
            keep = res["score"] >= 0.4
            res["_zncc_kept"] = O.zncc_batch(r, m, res["x0"][keep], res["y0"][keep], res["dx"][keep], res["dy"][keep])
            n = len(res["x0"])
        return time.perf_counter() - t0, n, res

    mon, ref = get_pair(0)
    S = mon.shape[0]
    one_pass(mon[:512], ref[:512], conf)                    # load / warm the library
    times, rows, gate_by_pair = [], [], {}
    live = None
    for k in range(max(1, runs)):
        b = k % n_pairs
        if k:
            mon, ref = get_pair(b)
        dt, n, res = one_pass(mon, ref, conf)
        times.append(dt)
        rows.append(n)
        if b not in gate_by_pair:
            gate_by_pair[b] = parity_gate((gpu_frames or {}).get(b), res, None if res is None else res.get("_zncc_kept"))
        if k == 0:
            if scored_frame is not None:
                fsp = full_scoring_gate(O, mon, ref, scored_frame)
            live = cv2_live(mon, ref, dict(maxCorners=conf.maxCorners), res)
            rows1 = max(256, S // 10)
            conf1 = O.default_conf(**dict(conf_kw, maxCorners=max(1, conf_kw["maxCorners"] * rows1 // S)))
            O.set_threads(1)
            t1 = sorted(one_pass(mon[:rows1], ref[:rows1], conf1)[0] for _ in range(3))[1]
            O.set_threads(cores)
        del res
    O.set_threads(min(O.max_threads(), O.team_size()))
    med = statistics.median(times)
    n = int(statistics.median(rows))
    out = {"value": S * S / 1e6 / med, "unit": "Mpx/s", "cores": cores, "kind": "port",
           "sample": f"{len(times)} full {S}x{S} pairs of the GPU loop ({len(gate_by_pair)} distinct), KLT + ZNCC, median pass of {min(times):.2f} .. {max(times):.2f} s, "
                     f"{cores} OpenMP threads ({os.cpu_count()} logical CPUs visible)",
           "keypoints_per_s": n / med, "seconds_per_pass": [round(t, 3) for t in times],
           "single_thread": {"value": rows1 * S / 1e6 / t1, "unit": "Mpx/s", "cores": 1,
                             "sample": f"top {rows1} rows of pair 0, maxCorners {conf1.maxCorners}, median of 3 passes, {t1:.2f} s"}}
    checked = [g for g in gate_by_pair.values() if g.get("checked")]
    par = {"checked": bool(checked), "pairs_gated": len(checked), "by_pair": {str(b): g for b, g in sorted(gate_by_pair.items())}}
    if checked:
        par["passed"] = all(g.get("passed") for g in checked)
        par["keypoints_identical_and_in_order"] = all(g.get("keypoints_identical_and_in_order") for g in checked)
        for key in ("max_abs_ddx_px", "max_abs_ddy_px", "max_abs_dscore", "max_abs_dzncc"):
            vals = [g[key] for g in checked if key in g]
            if vals:
                par[key] = max(vals)
    out["parity"] = par
    if scored_frame is not None:
        out["full_scoring_parity"] = fsp
    if live is not None:
        out["opencv_live"] = live
    return out


def cv2_live(mon, ref, conf_kw, oracle_res):
    """Only if OpenCV happens to be importable on the box (it is not part of the image): time the reference-equivalent
    sequence (`_to_uint8` -> cv2.Laplacian -> goodFeaturesToTrack -> 2x calcOpticalFlowPyrLK -> FB test, klt.py:83-172,
    407-436) and report how the oracle's key points compare - the true reference arithmetic."""
    try:
        import cv2
    except Exception:
        return None
    from oracle import oracle as O
    t0 = time.perf_counter()
    lap = [cv2.Laplacian(O.to_uint8(x), cv2.CV_8U, ksize=7) for x in (ref, mon)]
    mask = ((mon != 0) & (ref != 0)).astype(np.uint8)
    p0 = cv2.goodFeaturesToTrack(lap[0], mask=mask, maxCorners=conf_kw["maxCorners"], qualityLevel=0.1, minDistance=10, blockSize=15)
    lk = dict(winSize=(25, 25), maxLevel=1, criteria=(cv2.TERM_CRITERIA_EPS | cv2.TERM_CRITERIA_COUNT, 30, 0.03))
    p1, _, _ = cv2.calcOpticalFlowPyrLK(lap[0], lap[1], p0, None, **lk)
    p0r, _, _ = cv2.calcOpticalFlowPyrLK(lap[1], lap[0], p1, None, **lk)
    d = np.abs(p0 - p0r).reshape(-1, 2).max(-1)
    keep = d < np.float32(0.1)
    dt = time.perf_counter() - t0
    out = {"opencv": cv2.__version__, "threads": cv2.getNumThreads(), "seconds": dt, "Mpx_per_s": mon.size / 1e6 / dt, "matched": int(keep.sum())}
    if oracle_res is not None:
        mine = set(zip(oracle_res["x0"].astype(int).tolist(), oracle_res["y0"].astype(int).tolist()))
        theirs = set(map(tuple, p0.reshape(-1, 2)[keep].astype(int).tolist()))
        out["keypoints_in_common"] = len(mine & theirs)
        out["oracle_keypoints"] = len(mine)
    return out
