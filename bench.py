#!/usr/bin/env python3
"""Benchmark of the KARIOS matching hot path on MI355X (BASELINE.json metric).

`python bench.py --gpus N --steps K --warmup W` prints ONE JSON line.  With N > 1 and no launcher environment the process
SPAWNS N ranks itself (before anything touches the GPU), relays rank 0's line and fails if any rank fails; under an external
launcher (`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`) WORLD_SIZE must equal --gpus.

Headline (`value`): one "step" = one pass of the hot path over one synthetic Sentinel-2-sized pair that is already resident in
HBM: uint8 stretch -> Laplacian(k=7) -> auto mask -> Shi-Tomasi (GFTT) -> pyramidal LK forward/backward -> forward-backward
score -> per-key-point ZNCC of the rows with score >= 0.4 (BASELINE config 2: "Sentinel-2 10 m band pair (10980x10980), KLT
only, 1 MI355X"; default processing_configuration.json, i.e. one 10980^2 tile, maxCorners 20000), driven through the product's
`karios_amd.stream.FrameStream` (depth 2: two pairs queued on the one context behind the one being submitted - they execute one
after the other; the queue only absorbs host jitter).  With N > 1 every rank matches its own
band pair (weak scaling: the reference's tiles / bands are independent) and the per-band key-point blocks are all-gathered over
RCCL inside the timed region WITHOUT the host in the loop: the block goes from the library's stream straight into a send ring
in HBM, a side stream waits for it on the device and issues the all-gather asynchronously, counts are accumulated on the device
and read once, behind the last step (`karios_amd.parallel.RankBlockExchange`).  `KARIOS_BENCH_EXCHANGE=1` runs the same exchange
with a ONE-rank RCCL group at N = 1 (the code path of the driver's 8-GPU run, executed on one GPU: tests/test_gpu_bench.py).

Objects on the same line
  roofline      the largest kernel of the step: bytes it must move / its hipEvent span on the library's stream (see `roofline_of`);
  cpu_baseline  the CPU oracle (a port of the reference path) on the same pair, all usable cores + a 1-thread sample (rank 0,
                N = 1); its `parity` object is SURVEY 8(d)'s gate on the measured pair (GPU frame of the timed loop vs the oracle);
  end_to_end    the drop-in path: page-locked host rasters -> `karios_amd.matcher.KLT.match` -> DataFrame + ZNCC
                (PCIe-inclusive; never `value`);
  full_scoring  the WHOLE of `_handle_klt_results`' scoring (core.py:894-907) in the device call of the tile: KLT + ZNCC +
                `mutual_info_score` + `mi_score` (FrameStream(mutual_info=True)), with the MI kernel's roofline and - in the
                cpu_baseline leg - an in-run gate against the oracle on all rows of the measured pair;
  in_flight     the same workload with THREE independent pairs in flight on the one GPU (one library context each);
  config3       BASELINE config 3 (large-shift pre-alignment: phase correlation + shift_image + KLT) at 10980^2 with its own
                roofline, the path the transform took and a gate (offset == generator truth == oracle on a 1098^2 crop);
  config4       BASELINE config 4 as a FIXED workload (4 bands x tile_size 5490 = 16 units) split over the N ranks - strong
                scaling; each rank's units run over up to three library contexts (A/B against one in `contexts_in_flight_ab`);
  config5       BASELINE config 5 stand-in at 10980^2 (cross-sensor look + user mask), one GPU;
  oracle_sensitivity  precomputed (labelled): how far the two defensible roundings of the OpenCV-defined arithmetic move the result.
`--config 3` prints the config-3 object as the line of its own.
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic HBM bytes per pixel of one pair, per dense stage
STAGE_BYTES_PER_PX = {
    "minmax": 4.0,                   # read mon 2 + ref 2
    "stretch_laplacian_mask": 7.0,   # read 2+2, write lap_mon 1 + lap_ref 1 + mask 1
    "min_eigen": 5.0,                # read lap_ref 1, write eig 4            (two-kernel path only)
    "candidates": 5.0,               # read eig 4 + mask 1                    (two-kernel path only)
    "pyramid": 2.5,                  # read 1+1, write 1/4+1/4
}
FUSED_EIG_BYTES_PER_PX = 2.0         # fused K3+K4: read lap_ref 1 + mask 1; the eig map is never written (+ 8 B per emitted key)
LK_BYTES_PER_POINT = 6272.0          # 2 directions x 2 levels x (28x28 I-patch + 28x28 J-patch), u8
ZNCC_BYTES_PER_POINT = 7396.0        # 2 x 43x43 x 2 B
SELECT_BYTES_PER_CANDIDATE = 16.0    # candidate ranking
MI_BYTES_PER_POINT = 12996.0         # 2 x 57x57 x 2 B chips per scored key point
PHASE_BYTES_PER_PX_F64 = 116.0       # SURVEY 8(d) large-shift model executed in fp64 (reference precision)
PHASE_BYTES_PER_PX_F32 = 60.0        # SURVEY 8(d) large-shift model in float32 (28 forward + 12 cross power + 16 inverse + 4 arg-max)
SHIFT_BYTES_PER_PX = 4.0
HBM_PEAK_GBS = 8000.0                # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s measured streaming copy)
PMC_FILE = os.path.join("profiles", "pmc_traffic.json")
SENS_FILE = os.path.join("profiles", "r02_oracle_sensitivity.json")


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", type=int, default=2, choices=(2, 3), help="BASELINE config of the headline line")
    ap.add_argument("--size", type=int, default=10980, help="image side (BASELINE: 10980)")
    ap.add_argument("--share-gpu", action="store_true", help="development: the N ranks share the visible GPU(s) round-robin and the "
                    "collectives run on gloo / host memory (RCCL wants one device per rank)")
    ap.add_argument("--in-flight", action="store_true", help="(kept for compatibility: the in_flight object is part of the default line)")
    ap.add_argument("--no-in-flight", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true")
    ap.add_argument("--no-full-scoring", action="store_true")
    ap.add_argument("--no-config3", action="store_true")
    ap.add_argument("--no-config4", action="store_true")
    ap.add_argument("--no-config5", action="store_true")
    ap.add_argument("--no-one-pair", action="store_true", help="skip the one_pair_per_submission object (profiler runs: only the headline's kernels)")
    ap.add_argument("--no-sensitivity", action="store_true", help="skip the hard_content / tie_heavy / e2e_shape objects")
    ap.add_argument("--cpu-runs", type=int, default=3)
    ap.add_argument("--depth", type=int, default=2, help="units of the headline loop still pending when submit() returns (FrameStream depth; <= 2 on one context: "
                    "three frame slots).  Same throughput as 1 on a quiet box (1.082 against 1.083 ms, three runs each); a host hiccup of up to a step no longer idles the GPU")
    ap.add_argument("--pairs-per-submission", type=int, default=4, choices=(1, 2, 4),
                    help="band pairs of the headline loop that travel in ONE batched submission (FrameStream.submit_many -> km_klt_units_frame_submit: "
                         "one device pipeline for all of them; 4 = the 10 m bands of a Sentinel-2 product).  1: one pair per submission (rounds 1 - 4); "
                         "the line reports that loop too (`one_pair_per_submission`)")
    ap.add_argument("--timed-stage", default="auto", help="stage bracketed by HIP events inside the timed region (auto: the largest kernel; none)")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------------- launcher
def launch_ranks(a) -> int:
    """`python bench.py --gpus N` without a launcher: start N ranks of this script (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*
    in their environment), relay rank 0's JSON line, fail when any rank fails.  This parent never imports torch or touches
    HIP: a process that has initialised the GPU must not be replaced or forked on this pool."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), LOCAL_WORLD_SIZE=str(a.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), KARIOS_BENCH_LAUNCHER="bench.py (spawned ranks)")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=True if r == 0 else None))
    out0 = ""
    failed = None
    try:
        out0, _ = procs[0].communicate()
        for r, p in enumerate(procs):
            rc = p.wait()
            if rc != 0 and failed is None:
                failed = (r, rc)
    except BaseException:
        failed = failed or (-1, 1)
        raise
    finally:
        if failed is not None:
            for p in procs:                      # exactly the processes started above
                if p.poll() is None:
                    p.terminate()
    lines = [ln for ln in out0.splitlines() if ln.startswith("{")]
    for ln in out0.splitlines():
        if not ln.startswith("{"):
            print(ln, file=sys.stderr)
    if failed is not None:
        print(f"bench.py: rank {failed[0]} exited with status {failed[1]}", file=sys.stderr)
        return failed[1] or 1
    if not lines:
        print("bench.py: rank 0 printed no JSON line", file=sys.stderr)
        return 1
    print(lines[-1])
    return 0


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


def cpu_baseline(mon, ref, conf_kw, runs, gpu_frame=None, scored_frame=None):
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
    if scored_frame is not None:
        O.set_threads(cores)
        out["full_scoring_parity"] = full_scoring_gate(O, mon, ref, scored_frame)
        O.set_threads(min(O.max_threads(), O.team_size()))
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


# ---------------------------------------------------------------------------------------------------- full scoring
def full_scoring(ctx, pair, conf, S, steps):
    """The reference's per-tile loop scores every confident candidate three times (core.py:894-907: ZNCC, `mutual_info_score`,
    `mi_score`; the two mutual-information scores are ~90 % of its scoring time).  Here all three ride in the
    device call of the tile: one pair in flight through FrameStream(mutual_info=True), same pair as the headline."""
    from karios_amd.stream import FrameStream
    with FrameStream(0.4, depth=1, want_spans=True, mutual_info=True) as stream:
        last = [None]
        rows = [0]

        def take(res):
            for d in res:
                last[0] = d
                rows[0] += d.raw.n_rows

        for _ in range(6):
            take(stream.submit(pair, conf))
        take(stream.drain())
        ctx.sync()
        # three windows of `steps` pairs, the median window is the figure (a 20-step window is 25 ms: one host hiccup of a few
        # milliseconds - collector, scheduler - moved a single window from 1.16 to 1.33 ms on one box); all three are reported
        windows = []
        for _w in range(3):
            rows[0] = 0
            t0 = time.perf_counter()
            for _ in range(steps):
                take(stream.submit(pair, conf))
            take(stream.drain())
            ctx.sync()
            windows.append((time.perf_counter() - t0) / steps)
        dt = sorted(windows)[1]
        n_rows = rows[0] // steps
        # stage spans (untimed pass, every stage bracketed)
        ctx.set_profiling(True)
        ctx.set_option("profile_stage", -1)
        ctx.set_option("profile_every", 1)
        spans, n = {}, 0
        for _ in range(6):
            for d in stream.submit(pair, conf):
                if any(v > 0 for v in d.spans.values()):
                    n += 1
                    for k, v in d.spans.items():
                        spans[k] = spans.get(k, 0.0) + v
        for d in stream.drain():
            if any(v > 0 for v in d.spans.values()):
                n += 1
                for k, v in d.spans.items():
                    spans[k] = spans.get(k, 0.0) + v
        ctx.set_profiling(False)
    frame = last[0].frame
    n_scored = 0 if frame is None else int((frame["score"].to_numpy() >= np.float32(0.4)).sum())
    stage = {k: round(v / max(1, n), 4) for k, v in spans.items() if v > 0}
    mi_ms = stage.get("mutual_info", 0.0)
    roof = {"kernel": "mi_kernel (k_mi.hip): 32x32 joint histogram of two 57x57 chips per scored key point, both scores", "bound": "hbm",
            "bytes_model": f"{MI_BYTES_PER_POINT:.0f} B x {n_scored} scored key points (DESIGN section 4, K12)",
            "achieved": (MI_BYTES_PER_POINT * n_scored / (mi_ms * 1e-3) / 1e9) if mi_ms > 0 else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "span_ms": mi_ms or None}
    roof["frac"] = None if roof["achieved"] is None else roof["achieved"] / HBM_PEAK_GBS
    pmc = pmc_traffic("mi_kernel", S)
    roof["traffic"] = pmc.get("traffic")
    if pmc:
        roof.update({k: v for k, v in pmc.items() if k != "traffic"})
    return {"workload": f"BASELINE config 2 pair ({S}x{S}), KLT + ZNCC + mutual_info_score + mi_score = the whole scoring of _handle_klt_results "
                        "(api/core.py:894-907) in the tile's device call; one pair in flight (FrameStream(0.4, mutual_info=True))",
            "steps": steps, "ms_per_pair": dt * 1e3, "windows_ms_per_pair": [round(w * 1e3, 4) for w in windows], "Mpx_per_s": S * S / 1e6 / dt,
            "matched_keypoints_per_sec": n_rows / dt,
            "matched_keypoints_per_pair": n_rows, "scored_rows_per_pair": n_scored, "columns": (None if frame is None else list(frame.columns)),
            "stage_ms": stage, "roofline": roof}, frame


# ---------------------------------------------------------------------------------------------------- in flight
def in_flight(dev, conf, S, first_pair, n_ctx=3, pairs=60):
    """Throughput with `n_ctx` independent band pairs in flight on ONE GPU: one library context (stream + workspace) per pair,
    submitted round-robin through ONE `FrameStream`.  The latency-bound stretches of one pair (the corner-selection chain, the
    frame ordering, ZNCC) are filled by the dense stages of the others.  Reported next to the headline, whose timed region
    keeps ONE pair in flight so that its kernel durations - the roofline - are those of the kernels alone."""
    import torch
    from karios_amd import synth
    from karios_amd._lib import Context
    from karios_amd.resident import ResidentPair
    from karios_amd.stream import FrameStream
    data = [first_pair] + [synth.make_pair_torch(S, S, 0.5, 0.25, seed=20260101 + 10 * i, device=dev) for i in range(1, n_ctx)]
    torch.cuda.synchronize()
    ctxs = [Context(dev.index or 0) for _ in range(n_ctx)]
    prs = [ResidentPair.from_device_pointers(m.data_ptr(), r.data_ptr(), np.uint16, S, S, ctx=c, keepalive=(m, r)) for (m, r), c in zip(data, ctxs)]

    def run(stream, n):
        rows = 0
        for i in range(n):
            rows += sum(len(d.frame) for d in stream.submit(prs[i % n_ctx], conf) if d.frame is not None)
        rows += sum(len(d.frame) for d in stream.drain() if d.frame is not None)
        for c in ctxs:
            c.sync()
        return rows

    with FrameStream(0.4, depth=2 * n_ctx) as stream:
        run(stream, 4 * n_ctx)
        redone0 = stream.units_redone
        t0 = time.perf_counter()
        rows = run(stream, pairs)
        dt = time.perf_counter() - t0
        redone = stream.units_redone - redone0
    del prs, data
    for c in ctxs:
        c.close()
    return {"pairs_in_flight": n_ctx, "pairs": pairs, "ms_per_pair": dt / pairs * 1e3, "Mpx_per_s": S * S / 1e6 * pairs / dt,
            "matched_keypoints_per_sec": rows / dt, "tiles_redone": redone,
            "note": "independent pairs on separate library contexts (HIP streams) of one GPU through karios_amd.stream.FrameStream; "
                    "the headline value / roofline keep ONE context (its pairs execute one after the other)"}


# ---------------------------------------------------------------------------------------------------- config 4
def config4(dev, rank, world, coll_dev, steps, n_ctx_max=3, batched=False):
    """4 bands x tile_size 5490 = 16 work units of 10980^2 pairs (seeds 20260101 + 10 b), split round-robin over the ranks;
    every rank keeps only its units' regions (box + ZNCC halo) resident; a step = all 16 units + ONE all-gather of their blocks.
    `batched` (the object's value since round 5): a rank's units go through ONE batched submission (km_klt_units_frame_submit: one set of
    device launches for all of them, blocks straight into the send buffer at its row pitch); else unit by unit on `n_ctx_max` library
    contexts (units in flight fill each other's latency-bound stretches): the A/B in `contexts_in_flight_ab`."""
    import torch
    import torch.distributed as dist
    from karios_amd import synth
    from karios_amd._lib import Context
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
    n_ctx = 1 if batched else max(1, min(n_ctx_max, len(mine)))
    ctxs = [Context(dev.index or 0) for _ in range(n_ctx)]
    resident = []
    for b in sorted({u.band for u in mine}):
        mon_t, ref_t = synth.make_pair_torch(S, S, 0.5, 0.25, seed=20260101 + 10 * b, device=dev)
        for u in (u for u in mine if u.band == b):
            rx, ry = max(0, u.x_off - DEFAULT_HALO), max(0, u.y_off - DEFAULT_HALO)
            rw, rh = min(S, u.x_off + u.x_size + DEFAULT_HALO) - rx, min(S, u.y_off + u.y_size + DEFAULT_HALO) - ry
            m, r = mon_t[ry:ry + rh, rx:rx + rw].contiguous(), ref_t[ry:ry + rh, rx:rx + rw].contiguous()
            c = ctxs[len(resident) % n_ctx]
            pair = ResidentPair.from_device_pointers(m.data_ptr(), r.data_ptr(), np.uint16, rh, rw, ctx=c, keepalive=(m, r))
            pair.window = (rx, ry, S, S)
            resident.append((u, pair, (u.x_off - rx, u.y_off - ry, u.x_size, u.y_size)))
        del mon_t, ref_t
    torch.cuda.synchronize()
    for slot, (u, _, _) in enumerate(resident):
        send[slot, 0] = u.index
    torch.cuda.synchronize()

    redone = [0]

    def sink(c, slot):
        c.set_frame_sink(None if slot is None else send[slot, 1:].data_ptr(), 0 if slot is None else L * 4)

    from karios_amd.resident import submit_units

    class _one:                                        # (a unit of a batch behaves like a submitted tile for the repeat below)
        def __init__(self, batch, i):
            self.batch, self.i = batch, i

        def redo(self):
            self.batch.wait()
            return self.batch.redo(self.i)

    def step():
        pend = []
        if batched and resident:
            c0 = ctxs[0]
            for lo in range(0, len(resident), 16):
                chunk = resident[lo:lo + 16]
                c0.set_frame_sink(send[lo, 1:].data_ptr(), (len(chunk) - 1) * (1 + L) * 4 + L * 4, (1 + L) * 4)
                batch = submit_units([(pair, box, (u.x_off, u.y_off)) for u, pair, box in chunk], conf, 0.4)
                c0.set_frame_sink(None)
                if batch is None:
                    raise SystemExit("config 4: the batch form refused the units")
                pend += [_one(batch, i) for i in range(len(chunk))]
        else:
            for slot, (u, pair, box) in enumerate(resident):
                sink(pair.ctx, slot)
                pend.append(pair.submit_tile(conf, box=box, zncc_threshold=0.4, origin=(u.x_off, u.y_off)))
                sink(pair.ctx, None)
        for c in ctxs:
            c.sync()
        # a unit outside the fixed capacities of the sync-free corner path comes back flagged (header word 2): exact repeat, into the
        # same slot of the send buffer (never seen on a GPU of its own; two development ranks time-slicing ONE GPU do raise it)
        if resident:
            flags = send[:len(resident), 3].contiguous().view(torch.int32).cpu()
            again = [int(i) for i in torch.nonzero(flags).flatten()]
            for slot in again:
                raw = pend[slot].redo()          # (runs with the sink off: the repeated block is copied into the unit's slot here)
                send[slot, 1:1 + len(raw.block)] = torch.from_numpy(raw.block).to(dev)
                redone[0] += 1
            if again:
                for c in ctxs:
                    c.sync()
        if coll_dev.type == "cuda":
            blocks = gather_block_tensor(send, len(units))
        else:                                  # development: several gloo ranks share one GPU
            blocks = gather_block_tensor(send.cpu(), len(units))
        flagged = int((blocks[:, 2].contiguous().view(torch.int32) != 0).sum().item())
        if flagged:       # (cannot happen: flagged units were repeated through the exact path above)
            raise SystemExit(f"config 4: {flagged} unit(s) still flagged after the exact repeat")
        got = int((blocks[:, 1].contiguous().view(torch.int32) != 0).sum().item())      # header word 1 = Ninit: units that arrived
        return int(blocks[:, 0].contiguous().view(torch.int32).sum().item()), got

    def fence():
        for c in ctxs:
            c.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # ---- batched mode: the steps are PIPELINED like the headline's (FrameStream): step k + 1 is submitted before step k is collected - its
    # blocks arrive in the page-locked slot (header flags on the host: no read-back of the send buffer), flagged units are repeated,
    # the step's send buffer (two alternate) is gathered.  The device never waits for the host between steps.
    send2 = [send, send.clone()] if batched else None

    def submit_step(k):
        buf = send2[k % 2]
        out = []
        c0 = ctxs[0]
        for lo in range(0, len(resident), 16):
            chunk = resident[lo:lo + 16]
            c0.set_frame_sink(buf[lo, 1:].data_ptr(), (len(chunk) - 1) * (1 + L) * 4 + L * 4, (1 + L) * 4)
            batch = submit_units([(pair, box, (u.x_off, u.y_off)) for u, pair, box in chunk], conf, 0.4)
            c0.set_frame_sink(None)
            if batch is None:
                raise SystemExit("config 4: the batch form refused the units")
            out.append((lo, batch))
        return out

    def collect_step(k, batches):
        buf = send2[k % 2]
        for lo, batch in batches:
            for i, raw in enumerate(batch.wait()):           # (the blocks have left the device: the sink's copy is ahead of the host slot's)
                if raw.flags:
                    raw = batch.redo(i)
                    buf[lo + i, 1:1 + len(raw.block)] = torch.from_numpy(raw.block).to(dev)
                    redone[0] += 1
        blocks = gather_block_tensor(buf if coll_dev.type == "cuda" else buf.cpu(), len(units))
        hdr = blocks[:, :3].contiguous().view(torch.int32).cpu()             # ONE read-back per step: rows, Ninit, flags of every unit
        if int((hdr[:, 2] != 0).sum()):
            raise SystemExit("config 4: a unit is still flagged after the exact repeat")
        return int(hdr[:, 0].sum()), int((hdr[:, 1] != 0).sum())

    def run(n):
        if not batched:
            r = (0, 0)
            for _ in range(n):
                r = step()
            return r
        prev, r = None, (0, 0)
        for k in range(n):
            cur = submit_step(k) if resident else []
            if prev is not None:
                r = collect_step(k - 1, prev)
            prev = cur
        return collect_step(n - 1, prev)

    rows, got = run(2)
    fence()
    t0 = time.perf_counter()
    rows, got = run(steps)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=coll_dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    del resident
    for c in ctxs:
        c.close()
    return {"workload": "BASELINE config 4: 4 synthetic band pairs 10980x10980 uint16 (seeds 20260101+10b), tile_size 5490 -> 16 units, KLT + ZNCC, "
                        "each rank holds only its units' regions (box + 128 px halo); one all-gather of the 16 frame blocks per step",
            "scaling": "strong", "n_gpus": world, "units": len(units), "units_gathered": got,
            "units_per_rank": [len(units_of_rank(units, r, world)) for r in range(world)], "contexts_in_flight_per_rank": n_ctx,
            "submission": ("batched: one km_klt_units_frame_submit per rank and step, steps pipelined (step k + 1 submitted before step k is collected and gathered)"
                           if batched else f"unit by unit on {n_ctx} context(s), a host synchronisation per step (round 4's loop)"),
            "steps": steps, "ms_per_step": dt / steps * 1e3, "value": 4 * S * S / 1e6 / (dt / steps), "unit": "Mpx/s",
            "matched_keypoints_per_step": rows, "matched_keypoints_per_sec": rows / (dt / steps), "units_repeated_exactly_on_this_rank": redone[0]}


# ---------------------------------------------------------------------------------------------------- config 3
def config3_object(ctx, dev, S, steps, warmup, with_gate=True):
    """BASELINE config 3: the same pair shifted by (37.25, -20.75) px with --enable-large-shift-detection: phase correlation
    (LargeOffsetMatcher.match) -> integer shift_image -> KLT on the shifted pair -> offsets added back (core.py:233-252, 739-786)."""
    import torch
    from karios_amd import synth
    from karios_amd.core import KLTConfiguration
    from karios_amd.resident import ResidentPair
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

    for _ in range(max(1, warmup)):
        step()
    ctx.set_option("profile_stage", -1)
    ctx.set_profiling(True)
    ctx.sync()
    t0 = time.perf_counter()
    phase_ms = 0.0
    for _ in range(steps):
        off, frame, tp = step()
        phase_ms += tp
    ctx.sync()
    dt = time.perf_counter() - t0
    ctx.set_profiling(False)
    phase_ms /= steps
    # 60 B/px for a float32 transform, twice the FFT terms (116 B/px) when the transform runs in the reference's fp64 -
    # priced on the path the library actually took (km_phase_info: 1 = hand-written float32 FFT, 2 = fp64 fallback)
    path, margin = ctx.phase_info()
    algo = (PHASE_BYTES_PER_PX_F32 if path == 1 else PHASE_BYTES_PER_PX_F64) * S * S
    achieved = algo / (phase_ms * 1e-3) / 1e9
    kname = "phase_correlation_f32" if path == 1 else "phase_correlation_f64"
    out = {
        "workload": f"BASELINE config 3: synthetic Sentinel-2 pair {S}x{S} uint16 shifted by (37.25, -20.75) px, phase correlation -> "
                    "shift_image -> KLT (one tile, maxCorners 20000) -> offsets added back; inputs resident in HBM",
        "value": S * S / 1e6 / (dt / steps), "unit": "Mpx/s", "steps": steps, "ms_per_step": dt / steps * 1e3,
        "dtype": ("f32 FFT (integer shift accepted on a clear peak, fp64 otherwise)" if path == 1 else "f64 FFT (reference precision)")
                 + ", u8/int32 stencils, f32 LK solve",
        "detected_offset_row_col": [float(off[0]), float(off[1])],
        "matched_keypoints_per_pair": len(frame), "median_dx_dy": [float(np.median(frame["dx"])), float(np.median(frame["dy"]))],
        "stage_ms": {"phase_correlation": round(phase_ms, 3)},
        "phase_path": {"path": "float32 hand-written FFT" if path == 1 else "float64 hand-written FFT", "peak_margin": margin},
        "roofline": {"bound": "hbm", "kernel": "phase_correlation (2-D FFT of ref + i mon, cross-power, inverse 2-D FFT, arg-max)" if path == 1
                     else "phase_correlation (complex128: 2-D FFT of ref + i mon in place, cross-power, inverse 2-D FFT, arg-max)", "achieved": achieved,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, **pmc_traffic(kname, S),
                     "algorithmic_bytes_per_launch": algo, "kernel_ms": phase_ms},
    }
    # the same correlation in the reference's own arithmetic (complex128, k_fft64.hip - the path `phase_fp64`, unclear float32 peaks
    # and sides the float32 kernels do not factor take): timed beside the default path, same answer required
    if path == 1:
        ctx.set_option("phase_fp64", 1)
        try:
            off64 = pair.phase_offset()
            ctx.sync()
            n64 = max(2, min(steps, 5))
            t0 = time.perf_counter()
            for _ in range(n64):
                off64 = pair.phase_offset()
            ctx.sync()
            ms64 = (time.perf_counter() - t0) / n64 * 1e3
            p64, _ = ctx.phase_info()
        finally:
            ctx.set_option("phase_fp64", 0)
        a64 = PHASE_BYTES_PER_PX_F64 * S * S
        step64_ms = dt / steps * 1e3 - phase_ms + ms64
        out["phase_fp64"] = {"ms": round(ms64, 3), "path": "float64 hand-written FFT" if p64 == 2 else "?", "detected_offset_row_col": [float(off64[0]), float(off64[1])],
                             "equals_float32_path": bool(np.array_equal(off64, off)), "algorithmic_bytes": a64,
                             "roofline_frac": a64 / (ms64 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             # the figure to quote beside the reference (large_offset.py:39 computes in complex128): the same step with the
                             # float32 correlation's time replaced by the complex128 one's
                             "config3_ms_per_step_at_reference_precision": round(step64_ms, 3),
                             "config3_value_at_reference_precision_Mpx_s": round(S * S / 1e6 / (step64_ms * 1e-3), 1)}
    if with_gate:
        # gate: the detected offset equals the generator's truth, and - on a 1098^2 crop of the SAME pair, small enough for the
        # fp64 numpy oracle - the GPU's answer equals the oracle's
        from oracle import oracle as O
        c = min(S, 1098)
        mon_c, ref_c = mon_t[:c, :c].contiguous(), ref_t[:c, :c].contiguous()
        torch.cuda.synchronize()
        crop = ResidentPair.from_device_pointers(mon_c.data_ptr(), ref_c.data_ptr(), np.uint16, c, c, ctx=ctx, keepalive=(mon_c, ref_c))
        gpu_crop = crop.phase_offset()
        crop_path, _ = ctx.phase_info()
        ora_crop = O.phase_cross_correlation(mon_c.cpu().numpy().view(np.uint16), ref_c.cpu().numpy().view(np.uint16))
        truth = [-21.0, 37.0]
        out["gate"] = {"truth_row_col": truth, "full_size_equals_truth": [float(off[0]), float(off[1])] == truth,
                       "crop": c, "gpu_crop_row_col": [float(v) for v in gpu_crop], "oracle_crop_row_col": [float(v) for v in ora_crop],
                       "crop_path": "float32" if crop_path == 1 else "fp64",
                       "gpu_crop_equals_oracle": bool(np.array_equal(gpu_crop, ora_crop)),
                       "median_dx_dy_within_0.05_px": bool(abs(np.median(frame["dx"]) - 37.25) < 0.05 and abs(np.median(frame["dy"]) + 20.75) < 0.05)}
        g = out["gate"]
        g["fp64_equals_float32"] = bool(out.get("phase_fp64", {}).get("equals_float32_path", True))
        g["passed"] = bool(g["full_size_equals_truth"] and g["gpu_crop_equals_oracle"] and g["median_dx_dy_within_0.05_px"] and g["fp64_equals_float32"])
    return out


def config5_object(ctx, dev, S, steps):
    """BASELINE config 5 stand-in at full size on ONE GPU: cross-sensor look (mon 3x3 block-averaged and nearest-upsampled, gamma 0.8
    radiometry, shift (0.4, -0.3)) with a user mask zeroing ~20 % of the pixels (the DEM is never read by the matcher)."""
    import torch
    from karios_amd import synth
    from karios_amd.core import KLTConfiguration
    from karios_amd.resident import ResidentPair
    from karios_amd.stream import FrameStream
    conf = KLTConfiguration()
    mon_t, ref_t, mask_t = synth.make_cross_sensor_pair_torch(S, S, device=dev)
    torch.cuda.synchronize()
    pair = ResidentPair.from_device_pointers(mon_t.data_ptr(), ref_t.data_ptr(), np.uint16, S, S, ctx=ctx, mask_ptr=mask_t.data_ptr(),
                                             keepalive=(mon_t, ref_t, mask_t))
    last = [None]

    def run(stream, n):
        rows = 0
        for _ in range(n):
            for d in stream.submit(pair, conf):
                rows += 0 if d.frame is None else len(d.frame)
                last[0] = d.frame if d.frame is not None else last[0]
        for d in stream.drain():
            rows += 0 if d.frame is None else len(d.frame)
            last[0] = d.frame if d.frame is not None else last[0]
        ctx.sync()
        return rows

    with FrameStream(0.4, depth=1) as stream:
        run(stream, 3)
        t0 = time.perf_counter()
        rows = run(stream, steps)
        dt = time.perf_counter() - t0
        redone = stream.units_redone
    f = last[0]
    masked = float((mask_t == 0).float().mean().item())
    return {"workload": f"BASELINE config 5 stand-in: {S}x{S} uint16 pair, monitored image with a 30 m look (3x3 block mean, nearest x3), gamma 0.8, "
                        "shift (0.4, -0.3) px, user mask, KLT + ZNCC on one GPU; inputs resident in HBM",
            "value": S * S / 1e6 / (dt / steps), "unit": "Mpx/s", "steps": steps, "ms_per_step": dt / steps * 1e3,
            "masked_fraction": round(masked, 4), "matched_keypoints_per_pair": rows // steps, "tiles_redone": redone,
            "median_dx_dy": None if f is None or not len(f) else [float(np.median(f["dx"])), float(np.median(f["dy"]))]}


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


def _stream_timing(ctx, pair, conf, S, steps, boxes=None, group=1):
    """ms per pair of `pair` through FrameStream (depth 2, ZNCC of the confident rows), median of three windows of `steps` pairs; the
    stage table from an untimed pass with every stage bracketed; flags the synchronisation-free corner path raised and units repeated.
    `group`: pairs per batched submission (the headline's --pairs-per-submission: the sensitivity workloads are timed in the headline's form)."""
    from karios_amd.stream import FrameStream
    boxes = boxes or [None]
    group = max(1, int(group))
    units_of_submission = [(pair, b, None) for _ in range(group) for b in boxes]
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
            ("hard_content", lambda: synth.make_hard_pair_torch(S, S, device=dev),
             "monitored image = smooth sub-pixel warp (0 .. 0.6 px on top of (0.5, 0.25)) of the reference texture, 55.5 % of an independent texture "
             "of the same spectrum mixed in, additive noise sigma 200 DN (karios_amd.synth.make_hard_pair_torch)"),
            ("tie_heavy", lambda: synth.make_tie_heavy_pair_torch(S, S, device=dev),
             "both rasters quantised to 6 grey levels (k = 7 Laplacian 99 % saturated) and periodic with 96 px: the candidate list consists of "
             "exact eigenvalue ties ordered by raster index only (karios_amd.synth.make_tie_heavy_pair_torch)")):
        mon_t, ref_t = make()
        torch.cuda.synchronize()
        pair = ResidentPair.from_device_pointers(mon_t.data_ptr(), ref_t.data_ptr(), np.uint16, S, S, ctx=ctx, keepalive=(mon_t, ref_t))
        o = {"workload": f"{S}x{S} uint16 pair resident in HBM, default configuration (one tile, k = 7, maxCorners 20000), KLT + ZNCC; " + note}
        o.update(_stream_timing(ctx, pair, conf, S, steps, group=group))     # (the headline's form: `group` pairs per batched submission)
        whole = pair.match_tile_raw(conf, zncc_threshold=0.4)            # blocking call: the library's diagnostics of the whole pair
        st = ctx.stats()
        o["whole_pair_blocking_call"] = {"path_flags": _bits(st.path_flags, KM_PATH_NAMES), "tie_rows_of_fused_eigen_pass": int(st.tie_rows),
                                         "n_candidates": int(st.n_candidates), "rows": whole.n_rows}
        o["gate"] = _band_gate(O, pair, mon_t, ref_t, conf, y0, rows)
        out[name] = o
        del pair, mon_t, ref_t
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


# ---------------------------------------------------------------------------------------------------- roofline helpers
def pmc_traffic(kernel: str, S: int) -> dict:
    """HBM bytes per launch of `kernel` from the committed PMC passes (profiles/pmc_traffic.json: separate --pmc runs, gfx950
    correction 2 * FETCH_SIZE + WRITE_SIZE).  PRECOMPUTED - measured with rocprofv3 on an earlier run of the same command, not by
    this process - and labelled so."""
    path = os.path.join(ROOT, PMC_FILE)
    try:
        db = json.load(open(path))
    except Exception:
        return {"traffic": None}
    ent = db.get(kernel)
    if not isinstance(ent, dict) or str(S) not in ent:
        return {"traffic": None}
    out = {"traffic": ent[str(S)], "traffic_source": f"precomputed: {PMC_FILE}"
           + (f" (measured at commit {ent['measured_at']})" if ent.get("measured_at") else " (round-1 counters)")}
    busy = (ent.get("_detail") or {}).get("valu_pipe_busy")
    if busy is not None:
        # what actually bounds the stencil / tracker kernels: instruction issue.  4 cycles per VALU wave-instruction over the 1024 SIMDs'
        # cycles of the launch, from the same precomputed passes
        out["valu_pipe_busy"] = busy
        out["valu_pipe_busy_note"] = "4 * SQ_INSTS_VALU / (1024 SIMDs * kernel cycles), kernels serialised by the counter run (precomputed)"
    return out


def roofline_of(stage_ms: dict, S: int, n_init: int, n_cand: int, n_zncc: int, timed_stage: str | None, minmax_early: bool = False) -> dict:
    """Every stage's bytes-it-must-move / span, and the object for the LARGEST one.  Dense stages: SURVEY 8(d)'s per-pixel figures;
    the fused minimum-eigenvalue + candidate kernel is priced on what IT moves (source 1 B/px + mask 1 B/px + 8 B per emitted key) -
    SURVEY's 10 B/px for the two unfused steps counts an eigenvalue-map round trip the fusion removed and is reported next to it as
    `unfused_model`; LK 6272 B per corner, ZNCC 7396 B per scored row, corner ranking + selection 16 B per candidate."""
    px = float(S) * S
    model = {"minmax": STAGE_BYTES_PER_PX["minmax"] * px, "stretch_laplacian_mask": STAGE_BYTES_PER_PX["stretch_laplacian_mask"] * px,
             "pyramid": STAGE_BYTES_PER_PX["pyramid"] * px, "lk_fwd_bwd": LK_BYTES_PER_POINT * n_init, "zncc": ZNCC_BYTES_PER_POINT * n_zncc}
    fused = stage_ms.get("candidates", 0) == 0 and stage_ms.get("min_eigen", 0) > 0
    eig_name = "min_eigen_candidates_fused" if fused else "min_eigen"
    if fused:
        model[eig_name] = FUSED_EIG_BYTES_PER_PX * px + 8.0 * n_cand
    else:
        model["min_eigen"] = STAGE_BYTES_PER_PX["min_eigen"] * px
        model["candidates"] = STAGE_BYTES_PER_PX["candidates"] * px
    ms = dict(stage_ms)
    if fused:
        ms[eig_name] = ms.pop("min_eigen")
        ms.pop("candidates", None)
    if "sort" in ms or "select" in ms:
        ms["rank_select"] = ms.pop("sort", 0.0) + ms.pop("select", 0.0)
        model["rank_select"] = SELECT_BYTES_PER_CANDIDATE * n_cand
    table = {}
    for k, b in model.items():
        t = ms.get(k, 0.0)
        if t > 0 and b > 0:
            table[k] = {"ms": round(t, 4), "bytes": b, "achieved": b / (t * 1e-3) / 1e9, "frac": b / (t * 1e-3) / 1e9 / HBM_PEAK_GBS}
    # the pyramids run on the library's second stream beside the fused eigenvalue pass (they fill what that issue-bound kernel leaves):
    # their span is stretched by the sharing and is not on the critical path - never the "largest kernel"
    if "pyramid" in table:
        table["pyramid"]["overlapped"] = "second stream, beside min_eigen: the span is stretched by the sharing (0.10 ms alone)"
    # likewise the min / max of a unit submitted behind another one (KM_PATH_MM_EARLY): second stream, beside the PREVIOUS unit's LK /
    # FB test / ZNCC - HBM-bound work under instruction-bound kernels; its span covers that whole window
    hidden = {"pyramid"}
    if minmax_early and "minmax" in table:
        table["minmax"]["overlapped"] = "second stream, beside the previous unit's LK .. ZNCC (0.083 ms alone at 5.8 TB/s); LK pays ~0.03 ms for the sharing"
        hidden.add("minmax")
    dom = max((k for k in table if k not in hidden), key=lambda k: table[k]["ms"])
    d = table[dom]
    out = {"bound": "hbm", "kernel": dom, "achieved": d["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": d["frac"],
           **pmc_traffic(dom, S), "algorithmic_bytes_per_launch": d["bytes"], "kernel_ms": d["ms"],
           "kernel_ms_source": ("HIP events over the timed steps" if dom == timed_stage else "HIP events over an untimed pass of the same loop")}
    t = out.get("traffic")
    if t:
        out["frac_of_measured_traffic"] = t / (d["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS      # what the kernel really moved (PMC, precomputed) / time / peak
    if fused and eig_name in table:
        unf = (STAGE_BYTES_PER_PX["min_eigen"] + STAGE_BYTES_PER_PX["candidates"]) * px
        e = table[eig_name]
        e["unfused_model"] = {"bytes": unf, "achieved": unf / (e["ms"] * 1e-3) / 1e9, "frac": unf / (e["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              "note": "SURVEY 8(d) P3 + P4 = 10 B/px: what the two unfused steps would move (eig map written and read back)"}
    out["kernels"] = table
    dense = [k for k in ("minmax", "stretch_laplacian_mask", eig_name, "candidates") if k in table]
    db, dm = sum(table[k]["bytes"] for k in dense), sum(table[k]["ms"] for k in dense if k not in hidden)
    if "pyramid" in table:
        db += table["pyramid"]["bytes"]            # (their bytes count, their time hides under the eigenvalue pass)
    out["dense_path"] = {"bytes": db, "ms": dm, "achieved": db / (dm * 1e-3) / 1e9, "frac": db / (dm * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "note": "minmax + stretch/Laplacian/mask + fused eigenvalue pass (+ the pyramids' bytes, hidden beside it"
                                 + ("; the min / max bytes likewise: hidden beside the previous unit's LK)" if "minmax" in hidden else ")")}
    tb = sum(v["bytes"] for v in table.values())
    tm = sum(v["ms"] for k, v in table.items() if k not in hidden)
    out["all_stages"] = {"bytes": tb, "ms_serial_sum": tm, "achieved": tb / (tm * 1e-3) / 1e9, "frac": tb / (tm * 1e-3) / 1e9 / HBM_PEAK_GBS}
    return out


# ---------------------------------------------------------------------------------------------------- main
def main():
    a = parse()
    if a.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        sys.exit(launch_ranks(a))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        print(f"bench.py: --gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks", file=sys.stderr)
        sys.exit(2)
    import torch
    import torch.distributed as dist

    n_dev = torch.cuda.device_count()          # (counting devices does not initialise the GPU on this image)
    if n_dev < 1:
        raise SystemExit("bench.py needs an MI355X: karios_amd has no CPU path")
    share = a.share_gpu or os.environ.get("KARIOS_BENCH_SHARE_GPU") == "1"
    if local_rank >= n_dev and not share:
        print(f"bench.py: rank {rank} wants GPU {local_rank} but only {n_dev} are visible (--share-gpu: development runs on fewer GPUs)", file=sys.stderr)
        sys.exit(3)
    dev_index = local_rank % n_dev
    backend = "gloo" if share else os.environ.get("KARIOS_BENCH_BACKEND", "nccl")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: karios_amd has no CPU path")
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    coll_dev = dev if backend == "nccl" else torch.device("cpu")
    ranks_seen, devices = 1, [{"rank": 0, "device": dev_index, "name": torch.cuda.get_device_name(dev_index)}]
    # KARIOS_BENCH_EXCHANGE=1: the N > 1 code path - process group, RCCL all-gather of the frame blocks, device-side counting - with a
    # group of ONE rank (what an 8-GPU job runs, executed on the one GPU a builder has; tests/test_gpu_bench.py)
    force_exchange = world == 1 and os.environ.get("KARIOS_BENCH_EXCHANGE") == "1"
    if force_exchange:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
    if world > 1 or force_exchange:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL ("nccl" on ROCm) in production; gloo lets the multi-rank logic be exercised with several ranks sharing one GPU
        # (development box) - the collectives then run on CPU tensors
        if backend == "nccl":
            # RCCL's version banner goes to STDOUT unless NCCL_DEBUG is NONE (tools/rccl_banner_probe.py, run 17: NONE is the only
            # setting of the ones tried that removes it; RCCL_LOG_LEVEL does not) - stdout is the JSON line's.  An explicit NCCL_DEBUG wins.
            os.environ.setdefault("NCCL_DEBUG", "NONE")
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        one = torch.ones(1, device=coll_dev, dtype=torch.int64)
        dist.all_reduce(one)
        ranks_seen = int(one.item())
        mine = torch.tensor([rank, dev_index], device=coll_dev, dtype=torch.int64)
        allr = torch.empty(2 * world, device=coll_dev, dtype=torch.int64)
        dist.all_gather_into_tensor(allr, mine)
        devices = [{"rank": int(r), "device": int(d)} for r, d in allr.view(world, 2).tolist()]
        if ranks_seen != world:
            raise SystemExit(f"bench.py: the all-reduce saw {ranks_seen} ranks, expected {world}")

    from karios_amd import synth
    from karios_amd._lib import Context
    from karios_amd.core import KLTConfiguration
    from karios_amd.parallel import RankBlockExchange
    from karios_amd.resident import ResidentPair
    from karios_amd.stream import FrameStream

    ctx = Context(dev_index)
    S = a.size
    if a.config == 3:
        if world > 1:
            raise SystemExit("config 3 (a global 2-D FFT) does not shard: replicas only, run it with --gpus 1")
        o = config3_object(ctx, dev, S, a.steps, a.warmup)
        line = {"metric": "Mpixels/sec (+ matched keypoints/sec), Sentinel-2 10980^2 pair, large-shift pre-alignment + KLT",
                "value": o["value"], "unit": "Mpx/s", "n_gpus": 1, "steps": a.steps, "warmup": a.warmup, "ms_per_step": o["ms_per_step"],
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": o["dtype"], "data": "synthetic",
                "config": {"workload": o["workload"], "pairs_per_step": 1}, "cpu_baseline": None}
        line.update({k: v for k, v in o.items() if k not in ("value", "unit", "steps", "ms_per_step", "dtype", "workload")})
        print(json.dumps(line))
        return

    conf = KLTConfiguration()  # processing_configuration.json defaults: one tile, k=7, maxCorners 20000
    t_gen = time.perf_counter()
    mon_t, ref_t = synth.make_pair_torch(S, S, 0.5, 0.25, seed=20260101 + 10 * rank, device=dev)
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t_gen
    pair = ResidentPair.from_device_pointers(mon_t.data_ptr(), ref_t.data_ptr(), np.uint16, S, S, ctx=ctx, keepalive=(mon_t, ref_t))

    # One step = one band pair through the whole hot path, driven by the product's FrameStream: the main thread SUBMITS the pair
    # (km_klt_tile_frame_submit returns when the pair's last kernel and the copy of its frame block are enqueued, so the next
    # pair's dense stages queue right behind them); the stream's worker thread waits for the block and runs the host half (frame
    # block -> pandas DataFrame + radial error / angle columns, numpy as in the reference) while the device already works on the
    # next pair.  Every frame is complete before the closing fence, so K timed steps are K finished pairs.
    stage_sum = {}
    totals = {"rows": 0, "frames": 0, "n_init": 0, "redone": 0, "redone_rows": 0, "last": None}
    # the path's only exchange step: one all-gather of every rank's key-point block per step.  RCCL: issued on a side stream
    # behind a DEVICE-side wait for the block, counted on the device, read once behind the last step - the submitting thread never
    # waits for a collective (round 3 staged the block through the host and read a count back in every step)
    G = max(1, int(a.pairs_per_submission))
    depth = max(0, min(2, a.depth))
    ex = RankBlockExchange(ctx, conf.maxCorners, True, device=coll_dev, halves=depth + 1) if (world > 1 or force_exchange) else None
    step_no = [0, 0]                    # units submitted / (gloo) units handed to the exchange
    pend_of = {}                        # RCCL exchange: first step of a submission -> (its pending frame / batch, units)
    group_size = {}                     # first step of a submission -> pairs it carries (its stage spans cover all of them)

    def submit_group(n):
        """`n` consecutive steps (band pairs) as ONE submission: n = 1 the single-unit entry point, else a batched submission."""
        on_gpu = ex is not None and ex.on_gpu
        k = step_no[0]
        step_no[0] += n
        group_size[k] = n
        if n == 1:
            if on_gpu:
                ex.arm(k)
                return stream.submit(pair, conf, tag=k, on_submitted=lambda pend, k=k: pend_of.__setitem__(k, (pend, 1)))
            return stream.submit(pair, conf, tag=k)
        if on_gpu:
            ex.arm_many(k, n)
        return stream.submit_many([(pair, None, None)] * n, conf, tags=list(range(k, k + n)),
                                  on_submitted=(lambda pend, _i, k=k, n=n: pend_of.__setitem__(k, (pend, n))) if on_gpu else None)

    def run_steps(n, marks=None):
        """n steps in groups of G pairs per submission (the last group may be smaller); `marks`: host time after every submission."""
        done = 0
        while done < n:
            g = min(G, n - done)
            if ex is not None and ex.on_gpu and g > 1:
                g = min(g, ex.batch - step_no[0] % ex.batch)       # (a batched submission fills slots of ONE group of the send ring)
            take(submit_group(g))
            done += g
            if marks is not None:
                marks.append((time.perf_counter(), g))

    def take(results):
        """Finished steps: their frames, and the exchange of their blocks - ISSUED here, when the step has been collected (its block
        reached the send slot in HBM long ago): the side stream's device-side wait is then satisfied at once.  Issued at submission time
        the wait sat in a hardware queue for the whole step, and the runtime maps more streams than it has hardware queues (4 by default)
        onto shared queues - whenever the side stream shared one with the library's compute or second stream the next unit stalled behind
        it (1.07 - 1.20 ms per step from run to run).  The host still never waits for a collective."""
        for d in results:
            n_rows = d.raw.n_rows
            if ex is not None and ex.on_gpu:
                if d.tag in pend_of:                       # the first step of a submission: the whole submission has been collected
                    pend, n_units = pend_of.pop(d.tag)
                    if n_units == 1:
                        ex.issue(d.tag, pend)
                    else:
                        ex.issue_many(d.tag, n_units, pend)
            elif ex is not None:
                ex.issue(step_no[1], host_block=d.raw.block)
                step_no[1] += 1
            if d.redone:
                totals["redone_rows"] += n_rows
            totals["rows"] += n_rows
            totals["frames"] += 1
            totals["n_init"] = int(d.raw.block[:4].view(np.int32)[1])
            totals["n_candidates"] = d.raw.n_candidates
            totals["redone"] += int(d.redone)
            totals["last"] = d.frame
            gs = group_size.pop(d.tag, 1)
            for k, v in d.spans.items():
                stage_sum[k] = stage_sum.get(k, 0.0) + v
            if any(v > 0 for v in d.spans.values()):
                totals["span_samples"] = totals.get("span_samples", 0) + 1
                totals["span_units"] = totals.get("span_units", 0) + gs         # (a batched submission's spans cover all its pairs)

    def fence():
        ctx.sync()
        torch.cuda.synchronize()
        if world > 1 or force_exchange:
            dist.barrier()
            torch.cuda.synchronize()

    stream = FrameStream(0.4, depth=depth, want_spans=True)
    run_steps(a.warmup)
    take(stream.drain())
    # settle (untimed, on top of the W warm-up steps): a fresh box ramps its clocks over the first few hundred milliseconds of
    # load - windows of 20 steps are repeated until two consecutive ones agree within 2 % (at least 1 s, at most 3 s of work: the first
    # process on a fresh box was still 5 - 8 % slow after 0.3 s)
    settle = {"windows": 0, "seconds": 0.0}
    t_settle, prev = time.perf_counter(), None
    while True:
        fence()
        t_w = time.perf_counter()
        run_steps(20)
        take(stream.drain())
        fence()
        cur = time.perf_counter() - t_w
        settle["windows"] += 1
        elapsed = time.perf_counter() - t_settle
        done = (prev is not None and abs(cur - prev) <= 0.02 * prev and elapsed >= 1.0) or elapsed >= 3.0
        if world > 1:                      # every rank must leave the loop in the same round (the fence is a barrier)
            flag = torch.tensor([1 if done else 0], device=coll_dev, dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            done = bool(flag.item()) or settle["windows"] >= 150
        if done:
            break
        prev = cur
    settle["seconds"] = round(time.perf_counter() - t_settle, 3)
    settle["last_window_ms_per_step"] = round(cur / 20 * 1e3, 4)
    # HIP events on the library stream bracket ONE stage inside the timed region - the one the roofline is quoted on: a timed span
    # is two event records, i.e. two points where consecutive kernels may not overlap, and timing all ten stages costs ~0.07 ms
    # per pair.  The stage is the largest of the warm-up's full table; the full stage table of the line comes from a second,
    # untimed pass of the same loop right after.
    stage_names = [ctx.lib.km_stage_name(i).decode() for i in range(16)]
    # the interpreter's cyclic garbage collector stays out of the timed steps: a generation-2 pass over the object graph of torch +
    # pandas takes 10 - 20 ms - a quarter of a 60-step region - whenever the frames' allocations happen to trigger it.  It is run
    # HERE, in front of the probe steps, not right in front of the timed region: a full collection walks the whole heap (the hot
    # interpreter paths leave the CPU caches) and leaves the GPU idle for tens of milliseconds - the first timed steps then ran 5 - 10 %
    # slow (step_spread.in_order_ms showed it); the probe steps bring both back under load.
    import gc
    gc.collect()
    gc.disable()
    ctx.set_option("profile_stage", -1)
    ctx.set_profiling(True)
    stage_sum.clear()
    run_steps(8)
    take(stream.drain())
    fence()
    probe = {k: v for k, v in stage_sum.items() if k in ("stretch_laplacian_mask", "min_eigen", "lk_fwd_bwd") and v > 0}
    timed_stage = max(probe, key=probe.get) if probe else "min_eigen"
    if a.timed_stage not in ("auto", "none"):
        timed_stage = a.timed_stage
    ctx.set_option("profile_stage", stage_names.index(timed_stage))
    ctx.set_option("profile_every", 4)            # the timed steps are SAMPLED: every fourth records the stage's two events
    if a.timed_stage == "none":
        ctx.set_profiling(False)
    # second settle (untimed), in the exact configuration of the timed steps: the collection above idles the GPU for tens of
    # milliseconds and the clocks (and the host's caches) need more than the eight probe steps to come back - the first timed steps of
    # a 20-step region otherwise run 10 - 100 % slow on some boxes (round 4: in_order_ms 2.3, 1.7, 1.7, 1.4 ... behind a 1.09-ms settle)
    t_s2, prev2, settle["post_gc_windows"] = time.perf_counter(), None, 0
    while True:
        fence()
        t_w = time.perf_counter()
        run_steps(12)
        take(stream.drain())
        fence()
        cur2 = time.perf_counter() - t_w
        settle["post_gc_windows"] += 1
        el2 = time.perf_counter() - t_s2
        done2 = (prev2 is not None and abs(cur2 - prev2) <= 0.02 * prev2 and el2 >= 0.15) or el2 >= 1.0
        if world > 1 or force_exchange:
            flag = torch.tensor([1 if done2 else 0], device=coll_dev, dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            done2 = bool(flag.item()) or settle["post_gc_windows"] >= 60
        if done2:
            break
        prev2 = cur2
    settle["post_gc_seconds"] = round(time.perf_counter() - t_s2, 3)
    settle["post_gc_last_window_ms_per_step"] = round(cur2 / 12 * 1e3, 4)
    stage_sum.clear()
    totals.update(rows=0, frames=0, redone=0, redone_rows=0, span_samples=0, span_units=0)
    if ex is not None:
        ex.finish()                    # (everything issued so far is accounted for ...)
        ex.reset_counts()              # ... and the counters restart with the timed region
    fence()
    cpu0 = (time.thread_time(), stream.worker_cpu_s, time.process_time())
    t0 = time.perf_counter()
    marks = [(t0, 0)]
    run_steps(a.steps, marks)
    take(stream.drain())              # the last pair's frame: part of the timed region
    exchange = None
    if ex is not None:
        # every rank's matched key points of the K steps, counted on the device from the GATHERED blocks; units the synchronisation-
        # free corner path flagged were repeated exactly by their owner: their rows travel in one closing all-reduce
        rows_gathered, flagged_blocks = ex.finish()
        redone_rows = 0
        if flagged_blocks:               # (every rank read the same gathered headers: all of them enter the collective, or none)
            extra = torch.tensor([totals["redone_rows"]], device=coll_dev, dtype=torch.int64)
            dist.all_reduce(extra)
            redone_rows = int(extra.item())
        exchange = {"backend": backend if world > 1 else "nccl (one-rank group, KARIOS_BENCH_EXCHANGE=1)", "blocks_in": "HBM (km_set_frame_sink -> send ring)" if ex.on_gpu else "host (gloo development run)",
                    "steps_per_collective": ex.batch, "host_waits_per_step": 0 if ex.on_gpu else "lagged (gloo)", "send_ring_slots": ex.slots,
                    "rows_from_gathered_blocks": rows_gathered, "flagged_blocks_gathered": flagged_blocks, "rows_of_exactly_repeated_units": redone_rows,
                    "steps_exchanged": a.steps}
    fence()
    dt = time.perf_counter() - t0
    # host CPU a rank spends per step: the submitting thread (library calls, exchange enqueues, collecting frames),
    # the stream's worker thread (block -> DataFrame), and the whole process (copy pool, runtime threads): eight ranks' worth must fit
    # the box's CPU quota (this pool grants 16 CPUs) or the host becomes the 8-GPU bottleneck before xGMI does
    host_cpu = {"submit_thread_ms_per_step": round((time.thread_time() - cpu0[0]) / a.steps * 1e3, 4),
                "worker_thread_ms_per_step": round((stream.worker_cpu_s - cpu0[1]) / a.steps * 1e3, 4),
                "process_ms_per_step": round((time.process_time() - cpu0[2]) / a.steps * 1e3, 4)}
    gc.enable()
    # (where the region's time went, step by step: `value` is the whole region; a single slow step - another tenant of the host, a
    # page fault - shows here as max >> median instead of hiding in the mean)
    gaps_in_order = [round(1e3 * (b[0] - a_[0]) / max(1, b[1]), 3) for a_, b in zip(marks, marks[1:])]
    gaps = sorted(gaps_in_order)
    step_spread = {"median_ms": round(gaps[len(gaps) // 2], 4), "p90_ms": round(gaps[min(len(gaps) - 1, int(0.9 * len(gaps)))], 4),
                   "max_ms": round(gaps[-1], 4), "drain_ms": round(1e3 * (dt - (marks[-1][0] - t0)), 4),
                   "in_order_ms": gaps_in_order,
                   "note": "host-side intervals between consecutive submissions inside the timed region, per PAIR (a submission carries "
                           f"{G} pair(s); one context)"}
    assert totals["frames"] == a.steps
    n_kp_total = totals["rows"] if exchange is None else exchange["rows_from_gathered_blocks"] + exchange["rows_of_exactly_repeated_units"]
    frame = last_frame = totals["last"]   # (the parity gate of the cpu_baseline leg compares it with the oracle's result for the same pair)
    timed_samples = totals.get("span_samples", 0)
    timed_ms = stage_sum.get(timed_stage, 0.0) / max(1, totals.get("span_units", 0))       # per PAIR (a launch serves the pairs of its submission)
    timed_launch_ms = stage_sum.get(timed_stage, 0.0) / max(1, timed_samples)
    redone_timed = totals["redone"]
    # untimed pass: every stage bracketed
    ctx.set_profiling(True)
    ctx.set_option("profile_stage", -1)
    ctx.set_option("profile_every", 1)
    stage_steps = max(G, min(a.steps, 12) // G * G)
    stage_sum.clear()
    keep = dict(totals)
    run_steps(stage_steps)
    take(stream.drain())
    fence()
    totals.update(keep)
    ctx.set_profiling(False)
    stream.close()
    if ex is not None:
        ex.finish()                    # (the untimed pass armed the frame sink again: gathered, sink off - the objects below size their own blocks)
    if world > 1:
        t = torch.tensor([dt], device=coll_dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / a.steps * 1e3
    mpx_per_s = world * S * S / 1e6 / (dt / a.steps)
    host_cpu_ranks = [host_cpu["process_ms_per_step"]]
    if world > 1:
        mine_cpu = torch.tensor([host_cpu["process_ms_per_step"]], device=coll_dev, dtype=torch.float64)
        all_cpu = torch.empty(world, device=coll_dev, dtype=torch.float64)
        dist.all_gather_into_tensor(all_cpu, mine_cpu)
        host_cpu_ranks = [round(float(v), 4) for v in all_cpu.tolist()]
    host_cpu["process_ms_per_step_per_rank"] = host_cpu_ranks
    host_cpu["cpus_this_process_may_use"] = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count()
    host_cpu["note"] = ("time.thread_time / time.process_time over the timed steps; process = every thread of the rank (submit, FrameStream worker, the "
                        "library's copy pool, runtime threads).  Sum over the ranks / ms_per_step = CPUs the job keeps busy")
    host_cpu["cpus_busy_all_ranks"] = round(sum(host_cpu_ranks) / ms_per_step, 3)
    stats = ctx.stats()
    mm_early = bool(int(stats.path_flags) & 32)      # KM_PATH_MM_EARLY: the last unit's min / max ran beside its predecessor's LK
    stats.n_init = totals["n_init"]       # asynchronous submissions: the count travels in the frame block's header

    out = None
    if rank == 0:
        stage_ms = {k: v / stage_steps for k, v in stage_sum.items()}
        if timed_ms > 0:
            stage_ms[timed_stage] = timed_ms      # the roofline kernel: its average over the TIMED region
        n_cand = int(totals.get("n_candidates", 0) or stats.n_candidates)
        n_zncc = 0 if frame is None else int((frame["score"].to_numpy() >= 0.4).sum())
        roof = roofline_of(stage_ms, S, int(stats.n_init), n_cand, n_zncc, "min_eigen_candidates_fused" if timed_stage == "min_eigen" else timed_stage,
                           minmax_early=mm_early)
        if G > 1:
            # a launch of the batched pipeline serves the G pairs of its submission: bytes and duration both scale by G, `achieved` is
            # bytes per launch / launch duration either way; `kernel_ms` (and the stage table) are quoted per PAIR
            roof["pairs_per_launch"] = G
            roof["launch_ms"] = round(timed_launch_ms, 4) if timed_launch_ms > 0 else round(roof["kernel_ms"] * G, 4)
            roof["algorithmic_bytes_per_launch"] = roof["algorithmic_bytes_per_launch"] * G
            if roof.get("traffic"):
                roof["traffic"] = roof["traffic"] * G
            roof["note"] = (f"one launch = the kernel's work for the {G} pairs of a batched submission (km_klt_units_frame_submit): achieved = "
                            "algorithmic_bytes_per_launch / launch_ms; kernel_ms and the `kernels` table are per pair")
        out = {
            "metric": "Mpixels/sec (+ matched keypoints/sec), Sentinel-2 10980^2 pair, KLT + ZNCC",
            "value": mpx_per_s, "unit": "Mpx/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8/int32 stencils, f32 LK solve, f64 stretch+ZNCC", "data": "synthetic",
            "config": {"workload": f"BASELINE config 2: synthetic Sentinel-2 10 m band pair {S}x{S} uint16, shift (0.5, 0.25) px, "
                                   "KLT only (Laplacian k=7, maxCorners 20000, one tile), ZNCC of rows with score>=0.4, "
                                   "inputs resident in HBM; host DataFrame stage of pair i overlaps the device stage of the pairs behind it "
                                   f"(karios_amd.stream.FrameStream, depth {depth}); a stream of independent band pairs, {G} pair(s) per "
                                   "submission" + (" (FrameStream.submit_many -> km_klt_units_frame_submit: ONE device pipeline for the pairs of a submission, e.g. the "
                                                   "four 10 m bands of a product; frames bit-identical to one pair per submission, which `one_pair_per_submission` times)"
                                                   if G > 1 else ""), "pairs_per_step": world, "pairs_per_submission": G,
                       "parallelism": f"{world} independent band pair(s), 1 per GPU" + (", RCCL all-gather of key-point frames" if world > 1 else "")},
            "world": world, "launcher": os.environ.get("KARIOS_BENCH_LAUNCHER", "external (torch.distributed.run)" if "WORLD_SIZE" in os.environ else "single process"),
            "backend": (backend if (world > 1 or force_exchange) else None), "rccl_ranks_seen": ranks_seen, "devices": devices, "exchange": exchange,
            "matched_keypoints_per_sec": n_kp_total / dt,
            "host_cpu_ms_per_step": host_cpu,
            "matched_keypoints_per_pair": (0 if frame is None else len(frame)),
            "n_init": int(stats.n_init), "n_candidates": n_cand,
            "speculative_tiles_redone": int(redone_timed),
            "median_dx_dy": (None if frame is None else [float(np.median(frame["dx"])), float(np.median(frame["dy"]))]),
            "stage_ms": {k: round(v, 4) for k, v in stage_ms.items()},
            "stage_ms_note": f"per PAIR.  {timed_stage}: HIP events on {timed_samples} submissions of the timed region (every fourth call: an event record is a "
                             "point where consecutive kernels may not overlap); the other stages: an untimed pass of "
                             f"{stage_steps} steps right after (bracketing every stage costs ~0.07 ms per pair)"
                             + ("; rank + select of a batched submission are one span (`select`)" if G > 1 else ""),
            "roofline": roof,
            "synth_seconds": round(t_gen, 2), "settle": settle, "step_spread": step_spread, "python_gc": "disabled during the timed steps (collected in front of the 8 untimed probe steps that precede them)",
        }
        sens = os.path.join(ROOT, SENS_FILE)
        if os.path.exists(sens):
            try:
                out["oracle_sensitivity"] = dict(json.load(open(sens)), source=f"precomputed: {SENS_FILE} (tools/oracle_sensitivity.py, round 2; "
                                                 "its `seconds` are that tool's run time, not this process's)")
            except Exception:
                pass

    solo = rank == 0 and world == 1
    host_pair = None
    if solo and not (a.no_cpu_baseline and a.no_end_to_end):
        host_pair = (mon_t.cpu().numpy().view(np.uint16), ref_t.cpu().numpy().view(np.uint16))
    if solo and G > 1 and not a.no_one_pair:
        # the loop of rounds 1 - 4 for continuity: ONE pair per submission (km_klt_tile_frame_submit), same pair, same stream depth
        one = _stream_timing(ctx, pair, conf, S, max(8, min(20, a.steps)))
        out["one_pair_per_submission"] = {k: one[k] for k in ("ms_per_pair", "windows_ms_per_pair", "Mpx_per_s", "matched_keypoints_per_sec", "stage_ms",
                                                                "lk_span_ms", "selection_span_ms", "units_repeated_exactly")}
        out["one_pair_per_submission"]["note"] = ("FrameStream.submit(pair) per step (rounds 1 - 4's headline loop): every unit pays its corner-selection chain, "
                                                  "its LK fill + drain and its frame launches alone")
    scored_frame = None
    if solo and not a.no_full_scoring:
        out["full_scoring"], scored_frame = full_scoring(ctx, pair, conf, S, max(6, min(20, a.steps)))
    del pair
    if solo and not a.no_in_flight:
        out["in_flight"] = in_flight(dev, conf, S, (mon_t, ref_t))
    if solo and not a.no_end_to_end:
        out["end_to_end"] = end_to_end(host_pair[0], host_pair[1], ctx, max(4, min(12, a.steps)))
    del mon_t, ref_t
    torch.cuda.empty_cache()
    if solo and not a.no_config3:
        out["config3"] = config3_object(ctx, dev, S, max(3, min(6, a.steps // 3)), 2)
        torch.cuda.empty_cache()
    if solo and not a.no_config5 and hasattr(synth, "make_cross_sensor_pair_torch"):
        out["config5"] = config5_object(ctx, dev, S, max(4, min(10, a.steps // 2)))
        torch.cuda.empty_cache()
    if solo and not a.no_sensitivity:
        out.update(sensitivity_objects(ctx, dev, S, max(8, min(24, a.steps)), group=G))
        torch.cuda.empty_cache()
    if not a.no_config4 and S == 10980:
        c4 = config4(dev, rank, world, coll_dev, max(3, min(8, a.steps // 3)), batched=True)
        if solo:
            # A/B: the rank's units through ONE batched submission against unit by unit on one / three library contexts
            c4_one = config4(dev, rank, world, coll_dev, max(3, min(8, a.steps // 3)), n_ctx_max=1)
            c4_three = config4(dev, rank, world, coll_dev, max(3, min(8, a.steps // 3)), n_ctx_max=3)
            c4["contexts_in_flight_ab"] = {"batched": {"ms_per_step": c4["ms_per_step"], "units_repeated_exactly": c4["units_repeated_exactly_on_this_rank"]},
                                           "1": {"ms_per_step": c4_one["ms_per_step"], "units_repeated_exactly": c4_one["units_repeated_exactly_on_this_rank"]},
                                           "3": {"ms_per_step": c4_three["ms_per_step"], "units_repeated_exactly": c4_three["units_repeated_exactly_on_this_rank"]},
                                           "note": "same 16 units, same box, back to back; the object's value is the batched run (round 4: 10.8 ms per 16 units on "
                                                   "three contexts, 12.2 on one)"}
        if rank == 0:
            out["config4"] = c4
    if solo and not a.no_cpu_baseline:      # reported baseline: rank 0 at N=1 only
        cb = cpu_baseline(host_pair[0], host_pair[1], dict(maxCorners=conf.maxCorners), a.cpu_runs, gpu_frame=last_frame, scored_frame=scored_frame)
        if "full_scoring" in out:
            out["full_scoring"]["parity"] = cb.pop("full_scoring_parity", {"checked": False})
        out["cpu_baseline"] = cb
        out["speedup_vs_cpu_port"] = mpx_per_s / cb["value"]
    elif rank == 0:
        out["cpu_baseline"] = None
    if world > 1 or force_exchange:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # RCCL writes a banner (versions, library path) through C stdio, which is flushed when the process exits - behind everything
        # Python printed.  Flush it out now, so that the JSON line is the LAST line of stdout.
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
