#!/usr/bin/env python3
"""Randomised GPU-vs-oracle parity sweep of the tile pipeline (run on the GPU box).

Every case draws an image size, pixel type, shift, no-data pattern, user mask, Laplacian kernel sizes, polarity and
tracker parameters from a seeded generator, runs the oracle (`oracle.klt_tile` + `zncc_batch`) and the library
(`ops.klt_tile` through the C ABI from host buffers and `ResidentPair.match_tile` on resident data) and compares:

  corners p0            bit-identical (same points, same strength order)
  p1, p0r               bit-identical (the parity gate of SURVEY 8d is 1e-3 px; we report the stricter result)
  frame x0,y0,dx,dy,score   bit-identical, same (x0, y0) order
  zncc_score            <= 1e-9 (fp64), same NaN pattern

    python tools/fuzz_parity.py --cases 200 --seed 1 [--max-size 700]

Prints one line per failing case and a summary; exit code 1 when anything differed.  `tests/test_gpu_fuzz.py` runs a
short fixed slice of the same generator inside the GPU suite.
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

DTYPES = (np.uint16, np.uint16, np.uint8, np.int16, np.float32)
KSIZES = tuple(int(v) for v in os.environ["KARIOS_FUZZ_KSIZES"].split(",")) if os.environ.get("KARIOS_FUZZ_KSIZES") else (1, 3, 5, 7, 7, 9, 11)   # (env: bias a sweep towards some kernel sizes)
BLOCKS = (3, 5, 7, 9, 15, 15, 4, 8, 21)
WINS = (9, 15, 21, 25, 25, 31)
MAXC = (0, 40, 500, 5000, 20000)
QUAL = (0.001, 0.01, 0.1, 0.1, 0.3)


def draw_case(seed: int, max_size: int = 700):
    """-> dict describing one case (all randomness from `seed`)."""
    rng = np.random.default_rng(1000003 * seed + 17)
    H = int(rng.integers(40, max_size + 1))
    W = int(rng.integers(40, max_size + 1))
    if rng.random() < 0.15:                      # narrow / flat tiles: fewer strips than one wave covers, odd tails
        if rng.random() < 0.5:
            W = int(rng.integers(33, 130))
        else:
            H = int(rng.integers(33, 90))
    case = dict(
        seed=seed, H=H, W=W, dtype=DTYPES[rng.integers(len(DTYPES))],
        sx=float(rng.uniform(-1.5, 1.5)), sy=float(rng.uniform(-1.5, 1.5)),
        wedge=bool(rng.random() < 0.3), user_mask=bool(rng.random() < 0.3),
        nodata_mon=(1.0 if rng.random() < 0.2 else None), nodata_ref=(None if rng.random() < 0.8 else 0.0),
        mon_k=int(KSIZES[rng.integers(len(KSIZES))]), ref_k=int(KSIZES[rng.integers(len(KSIZES))]),
        invert=bool(rng.random() < 0.25),
        blocksize=int(BLOCKS[rng.integers(len(BLOCKS))]), winsize=int(WINS[rng.integers(len(WINS))]),
        maxCorners=int(MAXC[rng.integers(len(MAXC))]), qualityLevel=float(QUAL[rng.integers(len(QUAL))]),
        minDistance=float(rng.choice([0.0, 1.0, 2.5, 5.0, 10.0, 10.0, 14.3])),
        noise=float(rng.choice([0.0, 15.0, 60.0])),
    )
    if rng.random() < 0.5:
        case["ref_k"] = case["mon_k"]
    case["box"] = None
    if rng.random() < 0.35 and H >= 80 and W >= 80:   # an inner tile of the resident pair (row-strided views on the host path)
        bx, by = int(rng.integers(40, W + 1)), int(rng.integers(40, H + 1))
        case["box"] = (int(rng.integers(0, W - bx + 1)), int(rng.integers(0, H - by + 1)), bx, by)
    return case


def make_inputs(case):
    from karios_amd import synth
    rng = np.random.default_rng(case["seed"] * 7919 + 3)
    mon, ref = synth.make_pair(case["H"], case["W"], case["sx"], case["sy"], seed=20260101 + case["seed"],
                               noise_sigma=case["noise"], nodata_wedge=case["wedge"])
    dt = case["dtype"]
    if dt is np.uint8:
        mon, ref = (mon >> 5).clip(0, 255).astype(np.uint8), (ref >> 5).clip(0, 255).astype(np.uint8)
    elif dt is np.int16:
        mon, ref = (mon.astype(np.int32) - 4000).astype(np.int16), (ref.astype(np.int32) - 4000).astype(np.int16)
        if case["wedge"]:                         # keep the no-data value 0 where the wedge was
            yy, xx = np.ogrid[:case["H"], :case["W"]]
            wedge = (xx + yy) < 0.45 * case["W"]
            mon[wedge] = 0
            ref[wedge] = 0
    elif dt is np.float32:
        mon, ref = mon.astype(np.float32) * np.float32(0.37), ref.astype(np.float32) * np.float32(0.37)
        if rng.random() < 0.5:                    # NaNs count as invalid pixels and are ignored by the stretch
            for a in (mon, ref):
                yy = rng.integers(0, case["H"], 20)
                xx = rng.integers(0, case["W"], 20)
                a[yy, xx] = np.nan
    mask = None
    if case["user_mask"]:
        mask = np.full((case["H"], case["W"]), 255, np.uint8)
        for _ in range(int(rng.integers(1, 6))):
            y0, x0 = int(rng.integers(0, case["H"])), int(rng.integers(0, case["W"]))
            mask[y0:y0 + int(rng.integers(5, 120)), x0:x0 + int(rng.integers(5, 120))] = 0
    return mon, ref, mask


KNOB_DEFAULTS = dict(key_cap=0, stage_cap=0, topk_factor=0, select_first=0, defer=1, fused_eig=1, speculative=1, eig3=1, lk2=1,
                     stash_cap=0, mm_early=1)
DECLINED = [0]    # units mode: submissions the batch form declined for a documented reason (not covered -> unit by unit)
PATHS_HIT = {}   # KM_PATH_* bit -> number of library calls that went through it (coverage report of --force-paths)


def draw_knobs(seed: int) -> dict:
    """Test knobs of km_set_option for one case: shrunken capacities that make the corner detector's retry paths (key-buffer
    regrow, stage-overflow fallback, second selection pass, prefix growth) run in EVERY case instead of by timing luck."""
    rng = np.random.default_rng(31 * seed + 7)
    return dict(key_cap=int(rng.choice([0, 48, 256, 2048])), stage_cap=int(rng.choice([0, 0, 3, 40, 200])),
                topk_factor=int(rng.choice([0, 1, 1, 2])), select_first=int(rng.choice([0, 8, 100, 1000])),
                defer=int(rng.choice([1, 1, 0])), fused_eig=int(rng.choice([1, 1, 1, 0])), speculative=int(rng.choice([1, 1, 0])), eig3=int(rng.choice([1, 1, 0])),
                # round 3: both LK forms, the scatter launch's second read
                lk2=int(rng.choice([1, 1, 0])), stash_cap=int(rng.choice([0, 0, 1, 16])),
                mm_early=int(rng.choice([1, 1, 0])))


class forced_paths:
    """Context manager: apply / restore the knobs on the thread's default context and tally the paths taken."""

    def __init__(self, knobs):
        self.knobs = knobs

    def __enter__(self):
        from karios_amd._lib import default_context
        self.ctx = default_context()
        for k, v in (self.knobs or {}).items():
            self.ctx.set_option(k, v)
        return self

    def tally(self):
        f = int(self.ctx.stats().path_flags)
        for bit in (1, 2, 4, 8, 16):
            if f & bit:
                PATHS_HIT[bit] = PATHS_HIT.get(bit, 0) + 1

    def __exit__(self, *exc):
        for k, v in KNOB_DEFAULTS.items():
            self.ctx.set_option(k, v)
        return False


LAST = {}   # arrays of the case that ran last (dumped by main() when it failed: post-mortem of rare mismatches)


def run_case(case, ops, O, ResidentPair):
    """-> list of failure strings (empty = parity)."""
    LAST.clear()
    mon, ref, mask = make_inputs(case)
    conf = O.default_conf(maxCorners=case["maxCorners"], blocksize=case["blocksize"], matching_winsize=case["winsize"],
                          qualityLevel=case["qualityLevel"], minDistance=case["minDistance"],
                          laplacian_kernel_size={"mon": case["mon_k"], "ref": case["ref_k"]},
                          laplacian_invert_polarity=case["invert"])
    fails = []
    box = case.get("box")
    x_off, y_off, bx, by = box if box is not None else (0, 0, case["W"], case["H"])
    sl = (slice(y_off, y_off + by), slice(x_off, x_off + bx))
    mon_b, ref_b, mask_b = mon[sl], ref[sl], (None if mask is None else mask[sl])      # views: rows keep the full stride
    exp = O.klt_tile(np.ascontiguousarray(mon_b), np.ascontiguousarray(ref_b), conf,
                     mask_box=None if mask_b is None else np.ascontiguousarray(mask_b), nodata_mon=case["nodata_mon"],
                     nodata_ref=case["nodata_ref"], x_off=x_off, y_off=y_off, invert_mon=case["invert"])
    with forced_paths(case.get("knobs") or {}) as fp:
        status, tracks = ops.klt_tile(ref_b, mon_b, conf, mask_box=mask_b, nodata_ref=case["nodata_ref"], nodata_mon=case["nodata_mon"],
                                      mon_ksize=case["mon_k"], ref_ksize=case["ref_k"], invert_mon=case["invert"])
        fp.tally()
        st = fp.ctx.stats()      # post-mortem of a failing case: what the blocking call saw (its min / max, counts, paths)
        LAST["tile_stats"] = np.array([st.valid_pixels, st.n_candidates, st.n_init, st.min_ref, st.max_ref, st.min_mon, st.max_mon,
                                       st.max_eig, st.path_flags], np.float64)
        pair = ResidentPair.upload(mon, ref, mask=mask)
        pair.no_data_mon, pair.no_data_ref = case["nodata_mon"], case["nodata_ref"]
        if case.get("async_ring"):
            # ring stress: 4..6 submissions without a wait in between (3 slots: the oldest blocks are overwritten), the last
            # one is the frame under test
            pend = [pair.submit_tile(conf, box=box, zncc_threshold=0.4) for _ in range(case["async_ring"])] if conf.maxCorners > 0 else []
            frame = pend[-1].result().to_frame() if pend else pair.match_tile(conf, box=box, zncc_threshold=0.4)
        else:
            frame = pair.match_tile(conf, box=box, zncc_threshold=0.4)
        fp.tally()
    if status == "ok":
        LAST.update(gpu_p0=tracks[0].copy(), gpu_p1=tracks[1].copy(), gpu_p0r=tracks[2].copy())
    if frame is not None:
        LAST.update({f"gpu_frame_{c}": frame[c].to_numpy() for c in frame.columns})
    if exp is not None:
        LAST.update({f"exp_{k}": np.asarray(v) for k, v in exp.items() if k in ("x0", "y0", "dx", "dy", "score", "Ninit")})
    if exp is None:
        if status == "ok":
            fails.append(f"oracle None, library returned {len(tracks[0])} points")
        if frame is not None:
            fails.append("oracle None, resident frame not None")
        return fails
    if status != "ok":
        return [f"oracle has {exp['Ninit']} corners, library status {status}"]
    p0e = O.good_features(exp["lap_ref"], exp["mask"], conf.maxCorners, conf.qualityLevel, conf.minDistance, conf.blocksize)
    p1e = O.pyr_lk(exp["lap_ref"], exp["lap_mon"], p0e, case["winsize"])
    p0re = O.pyr_lk(exp["lap_mon"], exp["lap_ref"], p1e, case["winsize"])
    LAST.update(exp_p0=p0e, exp_p1=p1e, exp_p0r=p0re)
    def post_mortem():
        # a rare, non-reproducing mismatch of the blocking call (rounds 1 and 3): record what that call saw and whether the same call
        # right afterwards agrees with the oracle
        ts = LAST.get("tile_stats")
        truth = [float(np.nanmin(ref_b)), float(np.nanmax(ref_b)), float(np.nanmin(mon_b)), float(np.nanmax(mon_b))]
        fails.append(f"blocking call saw min/max {None if ts is None else ts[3:7].tolist()} (numpy: {truth}), valid {None if ts is None else int(ts[0])} "
                     f"(mask {int(np.count_nonzero(exp['mask']))}), path flags {None if ts is None else int(ts[8])}")
        for rep in range(3):
            s2, t2 = ops.klt_tile(ref_b, mon_b, conf, mask_box=mask_b, nodata_ref=case["nodata_ref"], nodata_mon=case["nodata_mon"],
                                  mon_ksize=case["mon_k"], ref_ksize=case["ref_k"], invert_mon=case["invert"])
            fails.append(f"repeat {rep}: " + ("status " + s2 if s2 != "ok" else
                                              f"p0 {np.array_equal(t2[0], p0e)} p1 {np.array_equal(t2[1], p1e)} p0r {np.array_equal(t2[2], p0re)}"))
        LAST.update(exp_lap_ref=exp["lap_ref"], exp_lap_mon=exp["lap_mon"], exp_mask=exp["mask"])
    if tracks[0].shape != p0e.shape or not np.array_equal(tracks[0], p0e):
        fails.append(f"p0 differs ({len(tracks[0])} vs {len(p0e)} corners)")
        post_mortem()
        return fails
    for name, got, want in (("p1", tracks[1], p1e), ("p0r", tracks[2], p0re)):
        if not np.array_equal(got, want):
            fails.append(f"{name} not bit-identical: max |diff| {np.abs(got - want).max():.3g} at {int(np.abs(got - want).argmax()) // 2}")
    if fails:
        post_mortem()
    n_exp = len(exp["x0"])
    if frame is None:
        if n_exp:
            fails.append(f"resident frame None, oracle has {n_exp} rows")
        return fails
    if len(frame) != n_exp:
        fails.append(f"frame rows {len(frame)} vs {n_exp}")
        return fails
    for col in ("x0", "y0", "dx", "dy", "score"):
        if not np.array_equal(frame[col].to_numpy(), exp[col]):
            fails.append(f"frame column {col} differs: max |diff| {np.abs(frame[col].to_numpy() - exp[col]).max():.3g}")
    keep = exp["score"] >= np.float32(0.4)
    z = np.full(n_exp, np.nan)
    if keep.any():
        z[keep] = O.zncc_batch(ref, mon, exp["x0"][keep], exp["y0"][keep], exp["dx"][keep], exp["dy"][keep])
    got = frame["zncc_score"].to_numpy()
    if not np.array_equal(np.isnan(got), np.isnan(z)):
        fails.append("zncc NaN pattern differs")
    elif np.nanmax(np.abs(got - z), initial=0.0) > 1e-9:
        fails.append(f"zncc max |diff| {np.nanmax(np.abs(got - z)):.3g}")
    return fails


def run_aux_case(seed, ops, O, ResidentPair):
    """Entry points around the tile pipeline: phase correlation, shift_image, MI / NMI, DN filter, the outlier-filter
    branch of `match_tile`.  -> list of failure strings."""
    from karios_amd import synth
    rng = np.random.default_rng(424243 * seed + 5)
    fails = []
    H, W = int(rng.integers(64, 420)), int(rng.integers(64, 420))
    dt = (np.uint16, np.uint8, np.int16, np.float32)[rng.integers(4)]
    # --- phase correlation of an integer-shifted copy (random shape: odd sizes, large prime factors) ---------------
    base, _ = synth.make_pair(H + 120, W + 120, 0.0, 0.0, seed=777 + seed, noise_sigma=0.0)
    sy, sx = int(rng.integers(-40, 41)), int(rng.integers(-40, 41))
    a = base[60:60 + H, 60:60 + W]
    b = base[60 - sy:60 - sy + H, 60 - sx:60 - sx + W]
    if dt is np.uint8:
        a, b = (a >> 5).astype(np.uint8), (b >> 5).astype(np.uint8)
    else:
        a, b = a.astype(dt), b.astype(dt)
    got, want = ops.phase_cross_correlation(b, a), O.phase_cross_correlation(b, a)
    if not np.array_equal(got, want):
        fails.append(f"phase correlation {got} vs oracle {want} (true shift {sy},{sx}; {H}x{W} {np.dtype(dt).name})")
    # ... and in the reference's complex128 whatever the side lengths (k_fft64.hip: levels, prime levels, Bluestein; ends fused or not)
    from karios_amd._lib import default_context
    ctx = default_context()
    ctx.set_option("phase_fp64", 1)
    ctx.set_option("f64_plain", int(rng.integers(2)))
    try:
        got64 = ops.phase_cross_correlation(b, a)
    finally:
        ctx.set_option("phase_fp64", 0)
        ctx.set_option("f64_plain", 0)
    if not np.array_equal(got64, want):
        fails.append(f"phase correlation (float64 path) {got64} vs oracle {want} (true shift {sy},{sx}; {H}x{W} {np.dtype(dt).name})")
    # --- shift_image, also beyond the image ------------------------------------------------------------------------
    yo, xo = int(rng.integers(-H - 3, H + 4)), int(rng.integers(-W - 3, W + 4))
    if rng.random() < 0.7:
        yo, xo = int(rng.integers(-50, 51)), int(rng.integers(-50, 51))
    img = a if rng.random() < 0.5 else a.astype(np.float64)
    if not np.array_equal(ops.shift_image(img, yo, xo), O.shift_image(img, yo, xo)):
        fails.append(f"shift_image({yo},{xo}) differs ({H}x{W} {img.dtype})")
    # --- scores and filters on random key points, including out-of-range ones ---------------------------------------
    mon, ref = synth.make_pair(H, W, float(rng.uniform(-1, 1)), float(rng.uniform(-1, 1)), seed=99 + seed,
                               nodata_wedge=bool(rng.random() < 0.4))
    n = int(rng.integers(1, 300))
    x0 = rng.integers(-5, W + 5, n).astype(np.float32)
    y0 = rng.integers(-5, H + 5, n).astype(np.float32)
    dx = rng.uniform(-3, 3, n).astype(np.float32)
    dy = rng.uniform(-3, 3, n).astype(np.float32)
    if rng.random() < 0.3:                        # x.5 sums: Python's banker's rounding of the float32 sum
        dx[: n // 2] = np.float32(0.5)
        dy[: n // 2] = np.float32(-1.5)
    gz, wz = ops.zncc_batch(ref, mon, x0, y0, dx, dy), O.zncc_batch(ref, mon, x0, y0, dx, dy)
    if not np.array_equal(np.isnan(gz), np.isnan(wz)) or np.nanmax(np.abs(gz - wz), initial=0.0) > 1e-9:
        fails.append("zncc_batch differs on random key points")
    (gs, gn), (ws, wn) = ops.mi_batch(ref, mon, x0, y0, dx, dy), O.mi_batch(ref, mon, x0, y0, dx, dy)
    for nm, g, w in (("studholme", gs, ws), ("nmi", gn, wn)):
        if not np.array_equal(np.isnan(g), np.isnan(w)) or np.nanmax(np.abs(g - w), initial=0.0) > 1e-9:
            fails.append(f"mi_batch {nm} differs: {np.nanmax(np.abs(g - w), initial=0.0):.3g}")
    # --- tile with the iterative outlier filter (klt.py:52-71) ------------------------------------------------------
    conf = O.default_conf(maxCorners=int(rng.choice([200, 2000])), outliers_filtering=True,
                          laplacian_kernel_size=int(rng.choice([3, 5, 7])))
    exp = O.klt_tile(mon, ref, conf)
    frame = ResidentPair.upload(mon, ref).match_tile(conf)
    if (exp is None) != (frame is None):
        fails.append("outlier-filter tile: None mismatch")
    elif exp is not None:
        for col in ("x0", "y0", "dx", "dy", "score"):
            if len(frame) != len(exp[col]) or not np.array_equal(frame[col].to_numpy(), exp[col]):
                fails.append(f"outlier-filter tile: column {col} differs")
                break
    return fails


def run_units_case(seed, ops, O, ResidentPair):
    """Batched submissions (km_klt_units_frame_submit): 2 .. 16 units - boxes of one or two resident pairs of one pixel type, widths off
    the dword grid included - in ONE device pipeline; every unit's frame (after the exact repeat of a flagged unit, as FrameStream does)
    against the oracle on that box: rows, order, x0 / y0 / dx / dy / score bit-identical, ZNCC <= 1e-9 with the same NaN pattern."""
    from karios_amd import resident
    rng = np.random.default_rng(seed * 104729 + 11)
    dt = (np.uint16, np.uint16, np.uint8, np.int16, np.uint16, np.uint8, np.int16, np.float32)[rng.integers(8)]
    pairs = []
    for k in range(int(rng.integers(1, 3))):
        case = dict(seed=seed * 3 + k, H=int(rng.integers(140, 720)), W=int(rng.integers(540, 1100)), dtype=dt, sx=float(rng.uniform(-1.5, 1.5)),
                    sy=float(rng.uniform(-1.5, 1.5)), wedge=bool(rng.random() < 0.3), user_mask=False, noise=float(rng.choice([0.0, 15.0, 60.0])))
        mon, ref, _ = make_inputs(case)
        pairs.append((ResidentPair.upload(mon, ref), mon, ref))
    nodata_mon = 1.0 if rng.random() < 0.2 else None
    nodata_ref = None if rng.random() < 0.8 else 0.0
    for p, _, _ in pairs:
        p.no_data_mon, p.no_data_ref = nodata_mon, nodata_ref
    k_mon = int(rng.choice(KSIZES))
    k_ref = k_mon if rng.random() < 0.6 else int(rng.choice(KSIZES))
    invert = bool(rng.random() < 0.25)
    win = int(WINS[rng.integers(len(WINS))])
    conf = O.default_conf(maxCorners=int(rng.choice([40, 500, 5000, 20000])), blocksize=int(rng.choice([3, 5, 7, 9, 15, 15])), matching_winsize=win,
                          qualityLevel=float(QUAL[rng.integers(len(QUAL))]), minDistance=float(rng.choice([0.0, 1.0, 2.5, 5.0, 10.0, 10.0, 10.0, 14.3])),
                          laplacian_kernel_size={"mon": k_mon, "ref": k_ref}, laplacian_invert_polarity=invert)
    units, meta = [], []
    for _ in range(int(rng.integers(2, 17))):
        pi = int(rng.integers(len(pairs)))
        pair, mon, ref = pairs[pi]
        H, W = mon.shape
        if rng.random() < 0.25:
            box = None
            x_off, y_off, bx, by = 0, 0, W, H
        else:
            bx = int(rng.integers(512, W + 1)); by = int(rng.integers(min(H, 70), H + 1))
            x_off = int(rng.integers(0, W - bx + 1)); y_off = int(rng.integers(0, H - by + 1))
            box = (x_off, y_off, bx, by)
        units.append((pair, box, None))
        meta.append((pi, x_off, y_off, bx, by))
    with forced_paths({}) as fp:
        pend = resident.submit_units(units, conf, zncc_threshold=0.4)
        if pend is None:
            # what the batch form documents as not covered (include/karios_hip.h): the caller goes unit by unit
            expected = (conf.minDistance < 1 or dt is np.float32 or
                        any(by < 2 * conf.blocksize + 8 or (bx + 1) // 2 <= win or (by + 1) // 2 <= win for _, _, _, bx, by in meta))
            DECLINED[0] += 1
            return [] if expected else [f"submit_units declined: {len(units)} units, kernels {k_mon}/{k_ref}, dtype {np.dtype(dt).name}, {meta}"]
        raws = pend.wait()
        fails = []
        for i, raw in enumerate(raws):
            if raw.flags:
                PATHS_HIT[16] = PATHS_HIT.get(16, 0) + 1
                raw = pend.redo(i)
            pi, x_off, y_off, bx, by = meta[i]
            _, mon, ref = pairs[pi]
            sl = (slice(y_off, y_off + by), slice(x_off, x_off + bx))
            exp = O.klt_tile(np.ascontiguousarray(mon[sl]), np.ascontiguousarray(ref[sl]), conf, mask_box=None, nodata_mon=nodata_mon, nodata_ref=nodata_ref,
                             x_off=x_off, y_off=y_off, invert_mon=invert)
            frame = raw.to_frame()
            n_exp = 0 if exp is None else len(exp["x0"])
            if frame is None or len(frame) == 0:
                if n_exp:
                    fails.append(f"unit {i} {meta[i]}: no frame, oracle has {n_exp} rows")
                continue
            if len(frame) != n_exp:
                fails.append(f"unit {i} {meta[i]}: rows {len(frame)} vs {n_exp}")
                continue
            for col in ("x0", "y0", "dx", "dy", "score"):
                if not np.array_equal(frame[col].to_numpy(), exp[col]):
                    fails.append(f"unit {i} {meta[i]}: column {col} differs, max |diff| {np.abs(frame[col].to_numpy() - exp[col]).max():.3g}")
            keep = exp["score"] >= np.float32(0.4)
            z = np.full(n_exp, np.nan)
            if keep.any():
                z[keep] = O.zncc_batch(ref, mon, exp["x0"][keep], exp["y0"][keep], exp["dx"][keep], exp["dy"][keep])
            got = frame["zncc_score"].to_numpy()
            if not np.array_equal(np.isnan(got), np.isnan(z)):
                fails.append(f"unit {i} {meta[i]}: zncc NaN pattern differs")
            elif np.nanmax(np.abs(got - z), initial=0.0) > 1e-9:
                fails.append(f"unit {i} {meta[i]}: zncc max |diff| {np.nanmax(np.abs(got - z)):.3g}")
    return fails


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--what", choices=("tile", "aux", "units"), default="tile")
    ap.add_argument("--cases", type=int, default=100)
    ap.add_argument("--seed", type=int, default=1, help="first case seed")
    ap.add_argument("--max-size", type=int, default=700)
    ap.add_argument("--budget-s", type=float, default=1e9, help="stop starting new cases after this many seconds")
    ap.add_argument("--force-paths", action="store_true", help="draw km_set_option test knobs per case so that the retry paths of the corner "
                    "detector (regrow, stage fallback, second pass, prefix growth) and the frame-ring wrap run in every case")
    a = ap.parse_args()
    from oracle import oracle as O
    O.build()
    from karios_amd import ops
    from karios_amd.resident import ResidentPair
    t0 = time.time()
    bad = done = 0
    for s in range(a.seed, a.seed + a.cases):
        if time.time() - t0 > a.budget_s:
            break
        case = draw_case(s, a.max_size) if a.what == "tile" else {a.what + "_seed": s}
        if a.force_paths and a.what == "tile":
            case["knobs"] = draw_knobs(s)
            case["async_ring"] = int(np.random.default_rng(s).integers(0, 7)) if s % 3 == 0 else 0
        try:
            fails = (run_case(case, ops, O, ResidentPair) if a.what == "tile" else run_aux_case(s, ops, O, ResidentPair) if a.what == "aux" else
                     run_units_case(s, ops, O, ResidentPair))
        except Exception as e:   # noqa: BLE001 - a crash in one case must not hide the others
            fails = [f"exception {type(e).__name__}: {e}"]
        done += 1
        if done % 5000 == 0:      # a killed soak still leaves its count behind
            print(f"progress: {done} cases, {bad} failing, {time.time() - t0:.0f} s", flush=True)
        if fails:
            bad += 1
            if LAST:
                os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                np.savez_compressed(os.path.join(ROOT, "gpurun_out", f"fuzz_fail_{a.what}_{s}.npz"), **LAST)
            print(f"FAIL seed {s}: {'; '.join(fails)}\n     {case}", flush=True)
    if True:
        names = {1: "key regrow", 2: "stage fallback", 4: "second selection pass", 8: "prefix growth", 16: "speculative tile repeated"}
        print("paths taken (library calls): " + ", ".join(f"{names[b]} {PATHS_HIT.get(b, 0)}" for b in (1, 2, 4, 8, 16)), flush=True)
    if a.what == "units":
        print(f"units mode: {DECLINED[0]} of {done} submissions declined by the batch form (documented limits), the others checked unit by unit against the oracle", flush=True)
    print(f"fuzz_parity: {done} cases (seeds {a.seed}..{a.seed + done - 1}), {bad} failing, {time.time() - t0:.1f} s", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
