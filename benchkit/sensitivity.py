"""The headline step on content that is not the best case (hard_content, tie_heavy, e2e_shape), each gated against the oracle in the run."""
from __future__ import annotations

import json
import os
import statistics
import time

import numpy as np

# ---------------------------------------------------------------------------------------------------- headline sensitivity
KM_FLAG_NAMES = {1: "shard_overflow", 2: "stage_overflow", 4: "kept_overflow", 8: "bin_too_large", 16: "cell_overflow", 32: "not_converged", 64: "slice_short"}
KM_PATH_NAMES = {1: "key_regrow", 2: "stage_fallback", 4: "second_pass", 8: "prefix_grown", 16: "spec_retry", 32: "mm_early"}


def _bits(v, names):
    return [n for b, n in names.items() if int(v) & b]


def _band_gate(O, pair, mon_t, ref_t, conf, y0, rows, x0=0, cols=None, with_iters=True, same_as=None):
    """In-run oracle gate on a box of the workload's own rasters (>= 1024 rows): the GPU's frame of that box (blocking tile call on the
    resident pair) against the oracle's for the same pixels - key points identical and in order, |d| <= 1e-3 px, ZNCC <= 1e-9 - plus,
    from the oracle on the same box, the forward-backward survival and the LK iteration histograms per level and direction."""
    S = pair.x_size
    cols = S - x0 if cols is None else cols
    box = (x0, y0, cols, rows)
    got = pair.match_tile(conf, box=box, zncc_threshold=0.4)
    st = pair.ctx.stats()
    mon_b = mon_t[y0:y0 + rows, x0:x0 + cols].cpu().numpy().view(np.uint16)
    ref_b = ref_t[y0:y0 + rows, x0:x0 + cols].cpu().numpy().view(np.uint16)
    oc = O.default_conf(maxCorners=conf.maxCorners, laplacian_kernel_size=conf.laplacian_kernel_size, tile_size=conf.tile_size)
    exp = O.klt_tile(mon_b, ref_b, oc, x_off=x0, y_off=y0)
    out = {"box_x_y_w_h": list(box), "rows_gpu": 0 if got is None else int(len(got)), "rows_oracle": 0 if exp is None else int(len(exp["x0"])),
           "path_flags": _bits(st.path_flags, KM_PATH_NAMES), "tie_rows": int(st.tie_rows), "n_candidates": int(st.n_candidates)}
    same = got is not None and exp is not None and len(got) == len(exp["x0"]) and bool(
        np.array_equal(got["x0"].to_numpy(), exp["x0"]) and np.array_equal(got["y0"].to_numpy(), exp["y0"]))
    out["keypoints_identical_and_in_order"] = bool(same)
    if same_as is not None:            # the frame another route produced for the same box (KLT.match): bit for bit the blocking call's
        out["frame_of_klt_match_identical"] = bool(got is not None and len(got) == len(same_as) and all(
            np.array_equal(got[c].to_numpy(), same_as[c].to_numpy()) for c in ("x0", "y0", "dx", "dy", "score")))
    if same:
        out["max_abs_ddx_px"] = float(np.abs(got["dx"].to_numpy() - exp["dx"]).max())
        out["max_abs_ddy_px"] = float(np.abs(got["dy"].to_numpy() - exp["dy"]).max())
        out["max_abs_dscore"] = float(np.abs(got["score"].to_numpy() - exp["score"]).max())
        keep = exp["score"] >= np.float32(0.4)
        # ZNCC chips are cut from the rasters the pair holds (the whole image), the oracle's from the same arrays
        mon_f = mon_t.cpu().numpy().view(np.uint16) if rows * cols < S * S else mon_b
        ref_f = ref_t.cpu().numpy().view(np.uint16) if rows * cols < S * S else ref_b
        zo = O.zncc_batch(ref_f, mon_f, exp["x0"][keep], exp["y0"][keep], exp["dx"][keep], exp["dy"][keep])
        zg = got["zncc_score"].to_numpy()[keep]
        out["zncc_nan_pattern_identical"] = bool(np.array_equal(np.isnan(zg), np.isnan(zo)))
        both = ~np.isnan(zg) & ~np.isnan(zo)
        out["max_abs_dzncc"] = float(np.abs(zg[both] - zo[both]).max()) if both.any() else 0.0
        out["passed"] = bool(out["max_abs_ddx_px"] <= 1e-3 and out["max_abs_ddy_px"] <= 1e-3 and out["max_abs_dscore"] <= 1e-2
                             and out["zncc_nan_pattern_identical"] and out["max_abs_dzncc"] <= 1e-9)
    else:
        out["passed"] = False
    if with_iters and exp is not None:
        p0 = O.good_features(exp["lap_ref"], exp["mask"], oc.maxCorners, oc.qualityLevel, oc.minDistance, oc.blocksize)
        p1, f0, f1 = O.pyr_lk(exp["lap_ref"], exp["lap_mon"], p0, oc.matching_winsize, return_iters="levels")
        p0r, b0, b1 = O.pyr_lk(exp["lap_mon"], exp["lap_ref"], p1, oc.matching_winsize, return_iters="levels")
        it = {}
        for name, a_ in (("forward_level1", f1), ("forward_level0", f0), ("backward_level1", b1), ("backward_level0", b0)):
            h = np.bincount(a_, minlength=31)[:31]
            it[name] = {"mean": round(float(a_.mean()), 3), "p90": int(np.percentile(a_, 90)), "at_cap_30": int(h[30]), "histogram_0_30": h.tolist()}
        it["mean_iterations_per_point_all_four"] = round(float(f0.mean() + f1.mean() + b0.mean() + b1.mean()), 3)
        out["lk_iterations_oracle_on_this_box"] = it
        out["forward_backward_survival_on_this_box"] = round(float(exp["Ninit"] and len(exp["x0"]) / exp["Ninit"]), 4)
    return out


def _stream_timing(ctx, pair, conf, S, steps, boxes=None, group=1, pairs=None):
    """ms per pair of `pair` through FrameStream (depth 2, ZNCC of the confident rows), median of three windows of `steps` pairs; the
    stage table from an untimed pass with every stage bracketed; flags the synchronisation-free corner path raised and units repeated.
    `group`: pairs per batched submission (the headline's form); `pairs`: the DISTINCT resident pairs that travel in one submission
    (pair i % len(pairs) is unit i; default: `pair` alone, e.g. the tiles of ONE raster pair)."""
    from karios_amd.stream import FrameStream
    boxes = boxes or [None]
    group = max(1, int(group))
    pairs = pairs or [pair]
    units_of_submission = [(pairs[g % len(pairs)], b, None) for g in range(group) for b in boxes]
    nsub = max(1, steps // group)
    steps = nsub * group

    def submit_one(stream):
        return stream.submit_many(units_of_submission, conf) if len(units_of_submission) > 1 else stream.submit(pair, conf, boxes[0])
    acc = {"rows": 0, "n_init": 0, "redone": 0, "units": 0, "flags": 0, "cand": 0}

    def take(res):
        for d in res:
            acc["rows"] += d.raw.n_rows
            acc["n_init"] += int(d.raw.block[:4].view(np.int32)[1])
            acc["cand"] += d.raw.n_candidates
            acc["redone"] += int(d.redone)
            acc["flags"] |= int(d.flags)
            acc["units"] += 1

    with FrameStream(0.4, depth=2, want_spans=True) as stream:
        for _ in range(4):
            take(submit_one(stream))
        take(stream.drain())
        ctx.sync()
        windows = []
        for _w in range(3):
            for k in acc:
                acc[k] = 0
            t0 = time.perf_counter()
            for _ in range(nsub):
                take(submit_one(stream))
            take(stream.drain())
            ctx.sync()
            windows.append((time.perf_counter() - t0) / steps)
        keep = dict(acc)
        ctx.set_profiling(True)
        ctx.set_option("profile_stage", -1)
        ctx.set_option("profile_every", 1)
        spans, n = {}, 0

        def take_spans(res):
            nonlocal n
            for d in res:
                if any(v > 0 for v in d.spans.values()):
                    n += 1
                    for k, v in d.spans.items():
                        spans[k] = spans.get(k, 0.0) + v

        for _ in range(4):
            take_spans(submit_one(stream))
        take_spans(stream.drain())
        ctx.set_profiling(False)
    dt = sorted(windows)[1]
    per_pair = max(1, n) * (group if len(units_of_submission) > 1 else 1)      # (a batched submission reports ONE set of spans for all its units)
    stage = {k: round(v / per_pair, 4) for k, v in spans.items() if v > 0}
    units = max(1, keep["units"])
    return {"ms_per_pair": dt * 1e3, "windows_ms_per_pair": [round(w * 1e3, 4) for w in windows], "Mpx_per_s": S * S / 1e6 / dt,
            "units_per_pair": len(boxes), "pairs_per_submission": group, "steps": steps,
            "corners_per_pair": keep["n_init"] // steps, "matched_keypoints_per_pair": keep["rows"] // steps,
            "forward_backward_survival": round(keep["rows"] / max(1, keep["n_init"]), 4), "candidates_per_pair": keep["cand"] // steps,
            "matched_keypoints_per_sec": keep["rows"] / steps / dt,
            "units_repeated_exactly": keep["redone"], "units_timed": units, "speculative_flags_seen": _bits(keep["flags"], KM_FLAG_NAMES),
            "stage_ms": stage, "lk_span_ms": stage.get("lk_fwd_bwd"), "selection_span_ms": round(stage.get("sort", 0.0) + stage.get("select", 0.0), 4)}


def sensitivity_objects(ctx, dev, S, steps, group=1):
    """the step on content that is NOT the best case, next to the headline (whose every corner survives the
    forward-backward test after ~2 LK iterations per level-0 pass).  Three resident 10980^2 workloads, each with ms per pair through
    FrameStream, the stage table, the flags / repeats of the synchronisation-free corner path and an in-run oracle gate on a box of
    >= 1024 rows: `hard_content` (about half of the tracks fail the round trip, like the reference's golden run: 37 448 of <= 80 000
    corners kept, tests/end_to_end/ref_data/test_full), `tie_heavy` (near-binary Laplacians of a periodic scene: exact eigenvalue ties),
    `e2e_shape` (the reference's end-to-end configuration, processing_configuration.json:8-18: tile_size 6000, Laplacian k = 5 -> four
    unequal tiles, through karios_amd.matcher.KLT.match on rasters resident in HBM)."""
    import torch
    from karios_amd import synth
    from karios_amd.core import KLTConfiguration
    from karios_amd.core.image import DeviceRasterImage
    from karios_amd.matcher import KLT
    from karios_amd.resident import ResidentPair
    from oracle import oracle as O
    O.set_threads(min(O.usable_cpus(), int(os.environ.get("KARIOS_ORACLE_THREADS", "1024"))))
    out = {}
    conf = KLTConfiguration()
    rows = min(S, 1024)
    y0 = max(0, (S - rows) // 2)
    for name, make, note in (
            ("hard_content", lambda seed: synth.make_hard_pair_torch(S, S, seed=seed, device=dev),
             "monitored image = smooth sub-pixel warp (0 .. 0.6 px on top of (0.5, 0.25)) of the reference texture, 55.5 % of an independent texture "
             "of the same spectrum mixed in, additive noise sigma 200 DN (karios_amd.synth.make_hard_pair_torch)"),
            ("tie_heavy", lambda seed: synth.make_tie_heavy_pair_torch(S, S, seed=seed, device=dev),
             "both rasters quantised to 6 grey levels (k = 7 Laplacian 99 % saturated) and periodic with 96 px: the candidate list consists of "
             "exact eigenvalue ties ordered by raster index only (karios_amd.synth.make_tie_heavy_pair_torch)")):
        data = [make(20260101 + 10 * b) for b in range(max(1, group))]      # DISTINCT pairs of a submission, like the headline's
        torch.cuda.synchronize()
        prs = [ResidentPair.from_device_pointers(m.data_ptr(), r.data_ptr(), np.uint16, S, S, ctx=ctx, keepalive=(m, r)) for m, r in data]
        (mon_t, ref_t), pair = data[0], prs[0]
        o = {"workload": f"{len(prs)} distinct {S}x{S} uint16 pairs resident in HBM, default configuration (one tile, k = 7, maxCorners 20000), KLT + ZNCC; " + note}
        o.update(_stream_timing(ctx, pair, conf, S, steps, group=group, pairs=prs))     # (the headline's form: `group` distinct pairs per batched submission)
        whole = pair.match_tile_raw(conf, zncc_threshold=0.4)            # blocking call: the library's diagnostics of the whole pair
        st = ctx.stats()
        o["whole_pair_blocking_call"] = {"path_flags": _bits(st.path_flags, KM_PATH_NAMES), "tie_rows_of_fused_eigen_pass": int(st.tie_rows),
                                         "n_candidates": int(st.n_candidates), "rows": whole.n_rows}
        o["gate"] = _band_gate(O, pair, mon_t, ref_t, conf, y0, rows)
        out[name] = o
        del pair, mon_t, ref_t, prs, data
        torch.cuda.empty_cache()
    # ---- the reference's end-to-end configuration on the headline's content
    conf_e = KLTConfiguration(tile_size=6000, laplacian_kernel_size=5)
    mon_t, ref_t = synth.make_pair_torch(S, S, 0.5, 0.25, device=dev)
    torch.cuda.synchronize()
    mon_img, ref_img = DeviceRasterImage(mon_t, np.uint16), DeviceRasterImage(ref_t, np.uint16)
    klt = KLT(conf_e, ctx=ctx)
    grid = klt.tile_boxes(S, S)
    frames = list(klt.match(mon_img, ref_img, None))
    ctx.sync()
    windows = []
    n_e = max(2, steps // 3)
    for _w in range(3):
        t0 = time.perf_counter()
        for _ in range(n_e):
            frames = list(klt.match(mon_img, ref_img, None))
        ctx.sync()
        windows.append((time.perf_counter() - t0) / n_e)
    dt = sorted(windows)[1]
    pair = ResidentPair.from_device_pointers(mon_t.data_ptr(), ref_t.data_ptr(), np.uint16, S, S, ctx=ctx, keepalive=(mon_t, ref_t))
    o = {"workload": f"{S}x{S} uint16 pair of the headline resident in HBM (karios_amd.core.DeviceRasterImage), the reference's end-to-end configuration "
                     "(tests/end_to_end/processing_configuration.json:8-18: tile_size 6000, laplacian_kernel_size 5, maxCorners 20000 per tile): "
                     f"{len(grid)} unequal tiles {[[t.x_size, t.y_size] for t in grid]} through karios_amd.matcher.KLT.match (bare frames, as KLT.match yields them)",
         "ms_per_pair": dt * 1e3, "windows_ms_per_pair": [round(w * 1e3, 4) for w in windows], "Mpx_per_s": S * S / 1e6 / dt, "tiles": len(grid),
         "matched_keypoints_per_pair": int(sum(len(f) for f in frames)), "matched_keypoints_per_sec": sum(len(f) for f in frames) / dt,
         "frames_yielded": len(frames)}
    st_e = _stream_timing(ctx, pair, conf_e, S, max(2, steps // 3), boxes=[tuple(t) for t in grid])
    o["with_zncc_through_framestream"] = {k: st_e[k] for k in ("ms_per_pair", "windows_ms_per_pair", "stage_ms", "lk_span_ms", "selection_span_ms",
                                                               "units_repeated_exactly", "speculative_flags_seen", "forward_backward_survival",
                                                               "corners_per_pair", "matched_keypoints_per_pair")}
    o["with_zncc_through_framestream"]["stage_ms_note"] = ("the four tiles of a pair as ONE batched submission (FrameStream.submit_many -> km_klt_units_frame_submit), "
                                                           "pairs pipelined (depth 2), ZNCC of the confident rows included; spans of the batch")
    t = grid[-1]                                                        # the smallest tile, whole: 4980 x 4980
    o["gate"] = _band_gate(O, pair, mon_t, ref_t, conf_e, t.y_off, t.y_size, t.x_off, t.x_size, with_iters=False, same_as=frames[-1])
    o["gate"]["passed"] = bool(o["gate"]["passed"] and o["gate"]["frame_of_klt_match_identical"] and len(frames) == len(grid))
    out["e2e_shape"] = o
    del pair, mon_t, ref_t, mon_img, ref_img
    torch.cuda.empty_cache()
    O.set_threads(min(O.max_threads(), O.team_size()))
    return out
