#!/usr/bin/env python3
"""Benchmark of the KARIOS matching hot path on MI355X (BASELINE.json metric).

Default (`--config 2`): one "step" = one pass of the hot path over one synthetic Sentinel-2-sized pair that is already
resident in HBM: uint8 stretch -> Laplacian(k=7) -> auto mask -> Shi-Tomasi (GFTT) -> pyramidal LK forward/backward ->
forward-backward score -> per-key-point ZNCC of the rows with score >= 0.4 (BASELINE config 2: "Sentinel-2 10 m band pair
(10980x10980), KLT only, 1 MI355X"; default processing_configuration.json, i.e. one 10980^2 tile, maxCorners 20000).
With N > 1 every rank matches its own band pair (weak scaling: the reference's tiles / bands are independent) and the
per-band key-point blocks are all-gathered over RCCL inside the timed region.

The same JSON line carries
  roofline      dominant dense kernel: algorithmic bytes / hipEvent stage time on the library's stream;
  cpu_baseline  the CPU oracle (a port of the reference path) on the same pair: median of 5 runs with all usable cores and a
                1-thread figure on a bounded sample (rank 0, N = 1 only); its `parity` object is SURVEY 8(d)'s gate on the measured
                pair: the GPU frame of the timed loop against the oracle's result (key points identical and in order, |d dx|, |d dy|
                <= 1e-3 px, score <= 1e-2, ZNCC <= 1e-9);
  end_to_end    the drop-in path a KARIOS user gets: page-locked host rasters -> `karios_amd.matcher.KLT.match` ->
                DataFrame + ZNCC column per pair, upload of pair i+1 under the compute of pair i (PCIe-inclusive; never `value`);
  in_flight     (with --in-flight) the same workload with THREE independent pairs in flight on the one GPU (a library context = stream + workspace
                each): the latency-bound stretches of one pair are filled by the others (N = 1 only).  The headline keeps one pair
                in flight so that the kernel durations behind `roofline` are those of the kernels alone;
  config4       BASELINE config 4 as a FIXED workload (4 bands x tile_size 5490 = 16 units) split over the N ranks -
                strong scaling; at N = 1 one GPU runs all 16 units;
  oracle_sensitivity  how far the two defensible roundings of the OpenCV-defined arithmetic can move the result
                (profiles/r02_oracle_sensitivity.json, produced by tools/oracle_sensitivity.py).

`--config 3` prints the line of BASELINE config 3 instead (large-shift pre-alignment: phase correlation + shift_image + KLT).
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic HBM bytes per pixel of one pair, per dense stage
STAGE_BYTES_PER_PX = {
    "minmax": 4.0,                   # read mon 2 + ref 2
    "stretch_laplacian_mask": 7.0,   # read 2+2, write lap_mon 1 + lap_ref 1 + mask 1
    "min_eigen": 5.0,                # read lap_ref 1, write eig 4
    "candidates": 5.0,               # read eig 4 + mask 1
    "pyramid": 2.5,                  # read 1+1, write 1/4+1/4
}
PHASE_BYTES_PER_PX_F64 = 116.0       # SURVEY 8(d) large-shift model executed in fp64 (reference precision)
PHASE_BYTES_PER_PX_F32 = 60.0        # SURVEY 8(d) large-shift model in float32 (28 forward + 12 cross power + 16 inverse + 4 arg-max)
SHIFT_BYTES_PER_PX = 4.0
HBM_PEAK_GBS = 8000.0                # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s measured streaming copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--in-flight", action="store_true", help="add the 3-pairs-in-flight throughput object (N=1 only; off by default: a "
                    "profile of the default command then shows every kernel running alone, like the events behind `roofline`)")
    ap.add_argument("--config", type=int, default=2, choices=(2, 3), help="BASELINE config of the headline line")
    ap.add_argument("--size", type=int, default=10980, help="image side (BASELINE: 10980)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true")
    ap.add_argument("--no-config4", action="store_true")
    ap.add_argument("--cpu-runs", type=int, default=5)
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------- CPU baseline
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


def cpu_baseline(mon, ref, conf_kw, runs, gpu_frame=None):
    """Oracle (kind 'port') on the SAME full pair, all usable cores: median of `runs` timed passes after one warm-up;
    plus a 1-thread figure on the top tenth of the image (maxCorners scaled to the same corner density)."""
    from oracle import oracle as O
    S = mon.shape[0]
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

    one_pass(mon[:512], ref[:512], conf)                    # load / warm the library
    times, n, res = [], 0, None
    for _ in range(max(1, runs)):
        dt, n, res = one_pass(mon, ref, conf)
        times.append(dt)
    med = statistics.median(times)
    rows1 = max(256, S // 10)
    conf1 = O.default_conf(**dict(conf_kw, maxCorners=max(1, conf_kw["maxCorners"] * rows1 // S)))
    O.set_threads(1)
    t1 = sorted(one_pass(mon[:rows1], ref[:rows1], conf1)[0] for _ in range(3))[1]
    O.set_threads(min(O.max_threads(), O.team_size()))
    out = {"value": S * S / 1e6 / med, "unit": "Mpx/s", "cores": cores, "kind": "port",
           "sample": f"the full {S}x{S} pair of the GPU run, KLT + ZNCC, median of {len(times)} passes ({min(times):.2f} .. {max(times):.2f} s), "
                     f"{n} matched key points; {cores} OpenMP threads = the CPUs this process may use ({os.cpu_count()} logical CPUs visible)",
           "keypoints_per_s": n / med,
           "single_thread": {"value": rows1 * S / 1e6 / t1, "unit": "Mpx/s", "cores": 1,
                             "sample": f"top {rows1} rows, maxCorners {conf1.maxCorners}, median of 3 passes, {t1:.2f} s"}}
    out["parity"] = parity_gate(gpu_frame, res, None if res is None else res.get("_zncc_kept"))
    live = cv2_live(mon, ref, dict(maxCorners=conf.maxCorners), res)
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


# ---------------------------------------------------------------------------------------------------- end to end
def end_to_end(mon, ref, ctx, steps):
    """What `KariosAPI._compute_matches` + `_handle_klt_results` would call (core.py:845-921): host rasters in, a scored
    DataFrame out, through the drop-in classes.  Two page-locked raster pairs alternate (GDAL would read into them); the next
    pair's upload is queued (`KLT.prefetch`) before the current pair is matched, so it travels under the compute."""
    from karios_amd import pinned_empty
    from karios_amd.core import KLTConfiguration, NumpyRasterImage
    from karios_amd.matcher import KLT, ZNCCService
    from karios_amd.resident import forget_shared_pairs
    conf = KLTConfiguration()
    pairs = []
    for k in range(2):
        pm, pr = pinned_empty(mon.shape, mon.dtype, ctx), pinned_empty(ref.shape, ref.dtype, ctx)
        np.copyto(pm, mon)
        np.copyto(pr, ref)
        pairs.append((NumpyRasterImage(pm), NumpyRasterImage(pr)))
    klt, zncc = KLT(conf, ctx=ctx), ZNCCService(ctx=ctx)

    def one(i):
        cur, nxt = pairs[i % 2], pairs[(i + 1) % 2]
        frames = klt.match(cur[0], cur[1], None)
        klt.prefetch(nxt[0], nxt[1], None)                  # queued BEFORE the generator runs: its copy overlaps this pair's kernels
        out = []
        for f in frames:
            dx, dy = f["dx"].to_numpy(), f["dy"].to_numpy()
            f["radial error"] = np.sqrt(dx ** 2 + dy ** 2)
            f["angle"] = np.degrees(np.arctan2(dy, dx))
            cand = f[f["score"] >= 0.4]
            f["zncc_score"] = zncc.compute_zncc(cand, cur[0], cur[1])
            out.append(f)
        return out

    klt.prefetch(*pairs[0], None)
    for i in range(2):
        one(i)
    t0 = time.perf_counter()
    rows = 0
    for i in range(steps):
        rows += sum(len(f) for f in one(2 + i))
    ctx.sync()
    dt = (time.perf_counter() - t0) / steps
    # the same loop from ordinary (pageable) numpy arrays: what an unmodified caller gets
    plain = (NumpyRasterImage(mon), NumpyRasterImage(ref))
    klt2 = KLT(conf, ctx=ctx)
    list(klt2.match(*plain, None))
    t1 = time.perf_counter()
    for _ in range(max(2, steps // 3)):
        for f in klt2.match(*plain, None):
            zncc.compute_zncc(f[f["score"] >= 0.4], *plain)
    dt_plain = (time.perf_counter() - t1) / max(2, steps // 3)
    klt._prefetched.clear()
    forget_shared_pairs()
    S = mon.shape[0]
    return {"ms_per_pair": dt * 1e3, "Mpx_per_s": S * S / 1e6 / dt, "keypoints_per_pair": rows // steps, "pairs": steps,
            "path": "page-locked host rasters (karios_amd.pinned_empty) -> KLT.match -> DataFrame + radial error / angle + ZNCCService.compute_zncc; "
                    "upload of pair i+1 (482 MB) on the copy stream under the compute of pair i",
            "upload_bytes_per_pair": int(mon.nbytes + ref.nbytes), "pcie_GBps": (mon.nbytes + ref.nbytes) / dt / 1e9,
            "pageable_numpy_ms_per_pair": dt_plain * 1e3}


# ---------------------------------------------------------------------------------------------------- config 4
def in_flight(dev, conf, S, n_ctx=3, pairs=48):
    """Throughput with `n_ctx` independent band pairs in flight on ONE GPU: one library context (stream + workspace) per pair,
    submitted round-robin by one thread, the host halves on one worker.  The latency-bound stretches of one pair (the corner-selection
    chain, the frame ordering, ZNCC) are filled by the dense stages of the others.  Reported next to the headline, whose timed region
    keeps ONE pair in flight so that its kernel durations - the roofline - are those of the kernels alone."""
    import torch
    from concurrent.futures import ThreadPoolExecutor
    from karios_amd import synth
    from karios_amd._lib import Context
    from karios_amd.resident import ResidentPair
    data = [synth.make_pair_torch(S, S, 0.5, 0.25, seed=20260101 + 10 * i, device=dev) for i in range(n_ctx)]
    torch.cuda.synchronize()
    ctxs = [Context(dev.index or 0) for _ in range(n_ctx)]
    prs = [ResidentPair.from_device_pointers(m.data_ptr(), r.data_ptr(), np.uint16, S, S, ctx=c, keepalive=(m, r)) for (m, r), c in zip(data, ctxs)]
    pool = ThreadPoolExecutor(max_workers=1)

    def host_half(pair, pend):
        raw = pend.wait()
        if raw.flags:
            return None
        return pair.score_frame(raw.to_frame(radial=True), 0.4)

    def run(n):
        futs, rows, redo = [], 0, 0
        def take(item):
            nonlocal rows, redo
            pair, pend, fut = item
            frame = fut.result()
            if frame is None:                    # flagged by the sync-free corner path: exact repeat on the submitting thread
                redo += 1
                frame = pair.score_frame(pend.redo().to_frame(radial=True), 0.4)
            rows += len(frame)
        for i in range(n):
            pr = prs[i % n_ctx]
            pend = pr.submit_tile(conf, zncc_threshold=0.4)
            futs.append((pr, pend, pool.submit(host_half, pr, pend)))
            if len(futs) > 2 * n_ctx:
                take(futs.pop(0))
        for item in futs:
            take(item)
        for c in ctxs:
            c.sync()
        return rows, redo

    run(4 * n_ctx)
    t0 = time.perf_counter()
    rows, redo = run(pairs)
    dt = time.perf_counter() - t0
    pool.shutdown()
    return {"pairs_in_flight": n_ctx, "pairs": pairs, "ms_per_pair": dt / pairs * 1e3, "Mpx_per_s": S * S / 1e6 * pairs / dt,
            "matched_keypoints_per_sec": rows / dt, "tiles_redone": redo,
            "note": "independent pairs on separate HIP streams of one GPU; the headline value / roofline keep one pair in flight"}


def config4(ctx, dev, rank, world, coll_dev, steps):
    """4 bands x tile_size 5490 = 16 work units of 10980^2 pairs (seeds 20260101 + 10 b), split round-robin over the ranks;
    every rank keeps only its units' regions (box + ZNCC halo) resident; a step = all 16 units + ONE all-gather of their blocks."""
    import torch
    import torch.distributed as dist
    from karios_amd import synth
    from karios_amd.core import KLTConfiguration
    from karios_amd.parallel import DEFAULT_HALO, block_len, enumerate_units, gather_block_tensor, units_of_rank
    from karios_amd.resident import ResidentPair
    S, conf = 10980, KLTConfiguration(tile_size=5490)
    units = enumerate_units(4, S, S, conf)
    mine = units_of_rank(units, rank, world)
    cap, L = conf.maxCorners, block_len(conf.maxCorners, True)
    per_rank = (len(units) + world - 1) // world
    send = torch.zeros((per_rank, 1 + L), dtype=torch.float32, device=dev)
    send[:, 0] = -1
    resident = []
    for b in sorted({u.band for u in mine}):
        mon_t, ref_t = synth.make_pair_torch(S, S, 0.5, 0.25, seed=20260101 + 10 * b, device=dev)
        for u in (u for u in mine if u.band == b):
            rx, ry = max(0, u.x_off - DEFAULT_HALO), max(0, u.y_off - DEFAULT_HALO)
            rw, rh = min(S, u.x_off + u.x_size + DEFAULT_HALO) - rx, min(S, u.y_off + u.y_size + DEFAULT_HALO) - ry
            m, r = mon_t[ry:ry + rh, rx:rx + rw].contiguous(), ref_t[ry:ry + rh, rx:rx + rw].contiguous()
            pair = ResidentPair.from_device_pointers(m.data_ptr(), r.data_ptr(), np.uint16, rh, rw, ctx=ctx, keepalive=(m, r))
            pair.window = (rx, ry, S, S)
            resident.append((u, pair, (u.x_off - rx, u.y_off - ry, u.x_size, u.y_size)))
        del mon_t, ref_t
    torch.cuda.synchronize()
    for slot, (u, _, _) in enumerate(resident):
        send[slot, 0] = u.index
    torch.cuda.synchronize()

    redone = [0]

    def step():
        pend = []
        for slot, (u, pair, box) in enumerate(resident):
            ctx.check(ctx.lib.km_set_frame_sink(ctx.handle, send[slot, 1:].data_ptr(), L * 4), "km_set_frame_sink")
            pend.append(pair.submit_tile(conf, box=box, zncc_threshold=0.4, origin=(u.x_off, u.y_off)))
        ctx.check(ctx.lib.km_set_frame_sink(ctx.handle, None, 0), "km_set_frame_sink")
        ctx.sync()
        # a unit outside the fixed capacities of the sync-free corner path comes back flagged (header word 2): exact repeat, into the
        # same slot of the send buffer (never seen on a GPU of its own; two development ranks time-slicing ONE GPU do raise it)
        if resident:
            flags = send[:len(resident), 3].contiguous().view(torch.int32).cpu()
            for slot in (int(i) for i in torch.nonzero(flags).flatten()):
                ctx.check(ctx.lib.km_set_frame_sink(ctx.handle, send[slot, 1:].data_ptr(), L * 4), "km_set_frame_sink")
                pend[slot].redo()
                ctx.check(ctx.lib.km_set_frame_sink(ctx.handle, None, 0), "km_set_frame_sink")
                redone[0] += 1
            if len(flags) and int(flags.count_nonzero()):
                ctx.sync()
        if coll_dev.type == "cuda":
            blocks = gather_block_tensor(send, len(units))
        else:                                  # development: several gloo ranks share one GPU
            blocks = gather_block_tensor(send.cpu(), len(units))
        flagged = int((blocks[:, 2].contiguous().view(torch.int32) != 0).sum().item())
        if flagged:       # (cannot happen: flagged units were repeated through the exact path above)
            raise SystemExit(f"config 4: {flagged} unit(s) still flagged after the exact repeat")
        return int(blocks[:, 0].contiguous().view(torch.int32).sum().item())

    def fence():
        ctx.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(2):
        rows = step()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        rows = step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=coll_dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    del resident
    return {"workload": "BASELINE config 4: 4 synthetic band pairs 10980x10980 uint16 (seeds 20260101+10b), tile_size 5490 -> 16 units, KLT + ZNCC, "
                        "each rank holds only its units' regions (box + 128 px halo); one all-gather of the 16 frame blocks per step",
            "scaling": "strong", "units": len(units), "units_per_rank": [len(units_of_rank(units, r, world)) for r in range(world)],
            "steps": steps, "ms_per_step": dt / steps * 1e3, "value": 4 * S * S / 1e6 / (dt / steps), "unit": "Mpx/s",
            "matched_keypoints_per_step": rows, "matched_keypoints_per_sec": rows / (dt / steps), "units_repeated_exactly_on_this_rank": redone[0]}


# ---------------------------------------------------------------------------------------------------- config 3
def config3_line(a, ctx, dev):
    """BASELINE config 3: the same pair shifted by (37.25, -20.75) px with --enable-large-shift-detection: phase correlation
    (LargeOffsetMatcher.match) -> integer shift_image -> KLT on the shifted pair -> offsets added back (core.py:233-252)."""
    import torch
    from karios_amd import synth
    from karios_amd.core import KLTConfiguration
    from karios_amd.resident import ResidentPair
    S = a.size
    conf = KLTConfiguration()
    mon_t, ref_t = synth.make_pair_torch(S, S, 37.25, -20.75, device=dev)
    torch.cuda.synchronize()
    pair = ResidentPair.from_device_pointers(mon_t.data_ptr(), ref_t.data_ptr(), np.uint16, S, S, ctx=ctx, keepalive=(mon_t, ref_t))

    def step():
        off = pair.phase_offset()                                  # [row, col]
        t_phase = ctx.stage_ms().get("phase_correlation", 0.0)
        shifted = pair.shifted_monitored(int(off[0]), int(off[1]))
        frame = shifted.match_tile(conf)
        frame["dx"] = frame["dx"] + np.float32(off[1])
        frame["dy"] = frame["dy"] + np.float32(off[0])
        return off, frame, t_phase

    for _ in range(max(1, a.warmup)):
        step()
    ctx.set_profiling(True)
    ctx.sync()
    t0 = time.perf_counter()
    phase_ms = 0.0
    for _ in range(a.steps):
        off, frame, tp = step()
        phase_ms += tp
    ctx.sync()
    dt = time.perf_counter() - t0
    ctx.set_profiling(False)
    phase_ms /= a.steps
    # 60 B/px for a float32 transform, twice the FFT terms (116 B/px) when the transform runs in the reference's fp64 -
    # priced on the path the library actually took (km_phase_info: 1 = hand-written float32 FFT, 2 = fp64 fallback)
    path, margin = ctx.phase_info()
    algo = (PHASE_BYTES_PER_PX_F32 if path == 1 else PHASE_BYTES_PER_PX_F64) * S * S
    achieved = algo / (phase_ms * 1e-3) / 1e9
    return {
        "metric": "Mpixels/sec (+ matched keypoints/sec), Sentinel-2 10980^2 pair, large-shift pre-alignment + KLT",
        "value": S * S / 1e6 / (dt / a.steps), "unit": "Mpx/s", "n_gpus": 1, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": ("f32 FFT (integer shift accepted on a clear peak, fp64 otherwise)" if path == 1 else "f64 FFT (reference precision)")
                 + ", u8/int32 stencils, f32 LK solve", "data": "synthetic",
        "config": {"workload": f"BASELINE config 3: synthetic Sentinel-2 pair {S}x{S} uint16 shifted by (37.25, -20.75) px, phase correlation -> "
                               "shift_image -> KLT (one tile, maxCorners 20000) -> offsets added back; inputs resident in HBM", "pairs_per_step": 1},
        "detected_offset_row_col": [float(off[0]), float(off[1])],
        "matched_keypoints_per_pair": len(frame), "median_dx_dy": [float(np.median(frame["dx"])), float(np.median(frame["dy"]))],
        "stage_ms": {"phase_correlation": round(phase_ms, 3)},
        "phase_path": {"path": "float32 hand-written FFT" if path == 1 else "fp64 rocFFT", "peak_margin": margin},
        "roofline": {"bound": "hbm", "kernel": "phase_correlation (2-D FFT of ref + i mon, cross-power, inverse 2-D FFT, arg-max)" if path == 1
                     else "phase_correlation (2x D2Z FFT, cross-power, Z2D FFT, arg-max)", "achieved": achieved,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "algorithmic_bytes_per_launch": algo, "kernel_ms": phase_ms},
        "cpu_baseline": None,
    }


# ---------------------------------------------------------------------------------------------------- main
def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: karios_amd has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    coll_dev = dev if os.environ.get("KARIOS_BENCH_BACKEND", "nccl") == "nccl" else torch.device("cpu")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL ("nccl" on ROCm) in production; KARIOS_BENCH_BACKEND=gloo lets the multi-rank logic be exercised with several
        # ranks sharing one GPU (development box) - the collectives then run on CPU tensors
        backend = os.environ.get("KARIOS_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from karios_amd import synth
    from karios_amd._lib import Context
    from karios_amd.core import KLTConfiguration
    from karios_amd.parallel import gather_rank_blocks
    from karios_amd.resident import ResidentPair

    ctx = Context(local_rank)
    if a.config == 3:
        if world > 1:
            raise SystemExit("config 3 (a global 2-D FFT) does not shard: replicas only, run it with --gpus 1")
        print(json.dumps(config3_line(a, ctx, dev)))
        return

    S = a.size
    conf = KLTConfiguration()  # processing_configuration.json defaults: one tile, k=7, maxCorners 20000
    t_gen = time.perf_counter()
    mon_t, ref_t = synth.make_pair_torch(S, S, 0.5, 0.25, seed=20260101 + 10 * rank, device=dev)
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t_gen
    pair = ResidentPair.from_device_pointers(mon_t.data_ptr(), ref_t.data_ptr(), np.uint16, S, S, ctx=ctx, keepalive=(mon_t, ref_t))

    # One step = one band pair through the whole hot path.  The main thread SUBMITS the pair (km_klt_tile_frame_submit: the
    # call returns when the pair's last kernel and the copy of its frame block are enqueued, so the next pair's dense
    # stages queue right behind them); a worker thread waits for the block and runs the host half (frame block -> pandas
    # DataFrame + radial error / angle columns, numpy as in the reference) while the device already works on the next
    # pair (ctypes releases the GIL inside the library).  Every frame is complete before the closing fence, so K timed
    # steps are K finished pairs.
    from concurrent.futures import ThreadPoolExecutor
    # the submitting thread now spends ~0.1 ms per pair inside the library (no host synchronisation in the corner path) and the
    # rest in Python next to the worker: with CPython's default 5 ms switch interval a thread that needs the GIL can wait that
    # long for the other to yield it
    sys.setswitchinterval(1e-4)
    pool = ThreadPoolExecutor(max_workers=1)
    stage_sum = {}
    totals = {"rows": 0, "frames": 0, "n_init": 0}

    def host_half(pend):
        raw = pend.wait()
        spans = pend.stage_ms()
        frame = None if raw.flags else raw.to_frame(radial=True)       # (radial error / angle built with the frame: one DataFrame construction)
        return pend, raw, spans, (None if frame is None else pair.score_frame(frame, 0.4))

    def collect(pending):
        """Result of an earlier step: its frame, and - the path's only exchange step - the all-gather of every rank's
        key-point block (device pipeline layout) over RCCL; the gathered blocks stay in HBM."""
        pend, raw, spans, frame = pending.result()
        if raw.flags:                       # the tile did not fit the synchronisation-free corner path: exact repeat (counted in the step)
            totals["redone"] = totals.get("redone", 0) + 1
            raw = pend.redo()
            frame = raw.to_frame()
            frame = None if frame is None else pair.score_frame(frame, 0.4)
        n_rows = raw.n_rows
        if world > 1:
            _, n_rows = gather_rank_blocks(raw.block, conf.maxCorners, True, device=coll_dev)
        totals["rows"] += n_rows
        totals["frames"] += 1
        totals["n_init"] = int(raw.block[:4].view(np.int32)[1])
        totals["n_candidates"] = raw.n_candidates
        for k, v in spans.items():
            stage_sum[k] = stage_sum.get(k, 0.0) + v
        return frame

    def step(pending):
        nxt = pool.submit(host_half, pair.submit_tile(conf, zncc_threshold=0.4))
        frame = collect(pending) if pending is not None else None
        return nxt, frame

    def fence():
        ctx.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    pending = None
    for _ in range(a.warmup):
        pending, _ = step(pending)
    if pending is not None:
        collect(pending)
        pending = None
    # HIP events on the library stream bracket ONE stage inside the timed region - the dominant dense kernel the roofline is
    # quoted on (the fused minimum-eigenvalue + candidate pass, stage "min_eigen"): a timed span is two event records, i.e. two
    # points where consecutive kernels may not overlap, and timing all ten stages costs ~0.07 ms per pair.  The full stage
    # table comes from a second, untimed pass of the same loop right after.
    eig_stage = [ctx.lib.km_stage_name(i).decode() for i in range(16)].index("min_eigen")
    ctx.set_option("profile_stage", eig_stage)
    ctx.set_profiling(True)
    stage_sum.clear()
    totals.update(rows=0, frames=0)
    fence()
    t0 = time.perf_counter()
    frame = None
    for _ in range(a.steps):
        pending, _ = step(pending)
    frame = collect(pending)          # the last pair's frame: part of the timed region
    fence()
    dt = time.perf_counter() - t0
    assert totals["frames"] == a.steps
    n_kp_total = totals["rows"]
    last_frame = frame                # (the parity gate of the cpu_baseline leg compares it with the oracle's result for the same pair)
    timed_eig_ms = stage_sum.get("min_eigen", 0.0) / a.steps
    # untimed pass: every stage bracketed
    ctx.set_option("profile_stage", -1)
    stage_steps = max(3, min(a.steps, 12))
    stage_sum.clear()
    keep = dict(totals)
    pending = None
    for _ in range(stage_steps):
        pending, _ = step(pending)
    collect(pending)
    fence()
    totals.update(keep)
    ctx.set_profiling(False)
    if world > 1:
        t = torch.tensor([dt], device=coll_dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / a.steps * 1e3
    mpx_per_s = world * S * S / 1e6 / (dt / a.steps)
    stats = ctx.stats()
    stats.n_init = totals["n_init"]       # asynchronous submissions: the count travels in the frame block's header

    out = None
    if rank == 0:
        stage_ms = {k: v / stage_steps for k, v in stage_sum.items()}
        if timed_eig_ms > 0:
            stage_ms["min_eigen"] = timed_eig_ms      # the roofline kernel: its average over the TIMED region
        bytes_per_px = dict(STAGE_BYTES_PER_PX)
        if stage_ms.get("candidates", 0) == 0 and stage_ms.get("min_eigen", 0) > 0:
            # default path: K3 + K4 fused in ONE kernel (no eig map) timed under "min_eigen"; the yardstick stays the
            # algorithmic figure of SURVEY 8(d) for the two steps it performs (P3 5 B/px + P4 5 B/px)
            bytes_per_px["min_eigen_candidates_fused"] = bytes_per_px.pop("min_eigen") + bytes_per_px.pop("candidates")
            stage_ms["min_eigen_candidates_fused"] = stage_ms.pop("min_eigen")
            stage_ms.pop("candidates", None)
        dense = {k: stage_ms[k] for k in bytes_per_px if stage_ms.get(k, 0) > 0}
        dom = max(dense, key=dense.get)
        # minmax runs as 2 launch pairs and the pyramid as 2 launches; the stage span is the unit that is timed
        algo_bytes = bytes_per_px[dom] * S * S
        achieved = algo_bytes / (dense[dom] * 1e-3) / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get(dom, {}).get(str(S))
            except Exception:
                traffic = None
        dense_ms = sum(dense.values())
        dense_bytes = sum(bytes_per_px[k] for k in dense) * S * S
        out = {
            "metric": "Mpixels/sec (+ matched keypoints/sec), Sentinel-2 10980^2 pair, KLT + ZNCC",
            "value": mpx_per_s, "unit": "Mpx/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8/int32 stencils, f32 LK solve, f64 stretch+ZNCC", "data": "synthetic",
            "config": {"workload": f"BASELINE config 2: synthetic Sentinel-2 10 m band pair {S}x{S} uint16, shift (0.5, 0.25) px, "
                                   "KLT only (Laplacian k=7, maxCorners 20000, one tile), ZNCC of rows with score>=0.4, "
                                   "inputs resident in HBM; host DataFrame stage of pair i overlaps the device stage of pair i+1", "pairs_per_step": world,
                       "parallelism": f"{world} independent band pair(s), 1 per GPU" + (", RCCL all-gather of key-point frames" if world > 1 else "")},
            "matched_keypoints_per_sec": n_kp_total / dt,
            "matched_keypoints_per_pair": (0 if frame is None else len(frame)),
            "n_init": int(stats.n_init), "n_candidates": int(totals.get("n_candidates", 0) or stats.n_candidates),
            "speculative_tiles_redone": int(totals.get("redone", 0)),
            "median_dx_dy": (None if frame is None else [float(np.median(frame["dx"])), float(np.median(frame["dy"]))]),
            "stage_ms": {k: round(v, 4) for k, v in stage_ms.items()},
            "stage_ms_note": f"min_eigen(_candidates_fused): HIP events over the {a.steps} timed steps; the other stages: an untimed pass of "
                             f"{stage_steps} steps right after (bracketing every stage costs ~0.07 ms per pair)",
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": algo_bytes, "kernel_ms": dense[dom],
                         "dense_path": {"bytes": dense_bytes, "ms": dense_ms, "achieved": dense_bytes / (dense_ms * 1e-3) / 1e9,
                                        "frac": dense_bytes / (dense_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}},
            "synth_seconds": round(t_gen, 2),
        }
        sens = os.path.join(ROOT, "profiles", "r02_oracle_sensitivity.json")
        if os.path.exists(sens):
            try:
                out["oracle_sensitivity"] = json.load(open(sens))
            except Exception:
                pass
    pool.shutdown()

    host_pair = None
    if rank == 0 and world == 1 and not (a.no_cpu_baseline and a.no_end_to_end):
        host_pair = (mon_t.cpu().numpy().view(np.uint16), ref_t.cpu().numpy().view(np.uint16))
    del pair
    if rank == 0 and world == 1 and not a.no_end_to_end:
        out["end_to_end"] = end_to_end(host_pair[0], host_pair[1], ctx, max(4, min(12, a.steps)))
    del mon_t, ref_t
    torch.cuda.empty_cache()
    if rank == 0 and world == 1 and a.in_flight:
        out["in_flight"] = in_flight(dev, conf, S)
        torch.cuda.empty_cache()
    if not a.no_config4 and S == 10980:
        c4 = config4(ctx, dev, rank, world, coll_dev, max(3, min(8, a.steps // 3)))
        if rank == 0:
            out["config4"] = c4
    if rank == 0 and world == 1 and not a.no_cpu_baseline:      # reported baseline: rank 0 at N=1 only
        cb = cpu_baseline(host_pair[0], host_pair[1], dict(maxCorners=conf.maxCorners), a.cpu_runs, gpu_frame=last_frame)
        out["cpu_baseline"] = cb
        out["speedup_vs_cpu_port"] = mpx_per_s / cb["value"]
    elif rank == 0:
        out["cpu_baseline"] = None
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
