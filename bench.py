#!/usr/bin/env python3
"""Benchmark of the KARIOS matching hot path on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over one synthetic Sentinel-2-sized pair that is already
resident in HBM: uint8 stretch -> Laplacian(k=7) -> auto mask -> Shi-Tomasi (GFTT) -> pyramidal LK
forward/backward -> forward-backward score -> per-key-point ZNCC of the rows with score >= 0.4
(BASELINE config 2: "Sentinel-2 10 m band pair (10980x10980), KLT only, 1 MI355X"; default
processing_configuration.json, i.e. one 10980^2 tile, maxCorners 20000).
With N > 1 every rank matches its own band pair (weak scaling; the reference's tiles / bands are
independent) and the per-band key-point frames are all-gathered over RCCL inside the timed region.

Prints ONE JSON line on rank 0.  `roofline` is computed from hipEvent stage times recorded on the
library's stream during the timed steps; `cpu_baseline` times the CPU oracle (oracle/, a port of the
reference path) on a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
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
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s measured streaming copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--size", type=int, default=10980, help="image side (BASELINE: 10980)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-rows", type=int, default=0, help="rows of the CPU sample (0 = half the image)")
    return ap.parse_args()


def cpu_baseline(mon_t, ref_t, size, conf_kw, sample_rows):
    """Oracle (kind 'port') on the top `rows` rows of the same pair, maxCorners scaled to keep the
    corner density of the full tile; all host cores via OpenMP."""
    from oracle import oracle as O
    rows = sample_rows if sample_rows > 0 else max(64, size // 2)
    rows = min(rows, size)
    mon = mon_t[:rows].cpu().numpy().view(np.uint16)
    ref = ref_t[:rows].cpu().numpy().view(np.uint16)
    frac = rows / size
    conf = O.default_conf(**dict(conf_kw, maxCorners=max(1, int(round(conf_kw["maxCorners"] * frac)))))
    cores = O.max_threads()      # OpenMP team = min(logical CPUs, affinity, cgroup CPU quota), see oracle.usable_cpus()
    O.klt_tile(mon[:256, :256], ref[:256, :256], conf)  # load / warm the library
    t0 = time.perf_counter()
    res = O.klt_tile(mon, ref, conf)
    n = 0
    if res is not None:
        keep = res["score"] >= 0.4
        O.zncc_batch(ref, mon, res["x0"][keep], res["y0"][keep], res["dx"][keep], res["dy"][keep])
        n = len(res["x0"])
    dt = time.perf_counter() - t0
    out = {"value": rows * size / 1e6 / dt, "unit": "Mpx/s", "cores": cores, "kind": "port",
           "sample": f"top {rows} rows x {size} cols of the same pair, maxCorners {conf.maxCorners} "
                     f"(same corner density), KLT + ZNCC, {dt:.2f} s, {n} matched key points; {cores} OpenMP threads = the CPUs this "
                     f"process may use ({os.cpu_count()} logical CPUs visible)",
           "keypoints_per_s": n / dt}
    live = cv2_live(mon, ref, dict(maxCorners=conf.maxCorners), res)
    if live is not None:
        out["opencv_live"] = live
    return out, res


def cv2_live(mon, ref, conf_kw, oracle_res):
    """Only if OpenCV happens to be importable on the box (it is not part of the image): time the reference-equivalent
    sequence (`_to_uint8` -> cv2.Laplacian -> goodFeaturesToTrack -> 2x calcOpticalFlowPyrLK -> FB test, klt.py:83-172,
    407-436) on the CPU sample and report how the oracle's key points compare - the true reference arithmetic."""
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

    S = a.size
    conf = KLTConfiguration()  # processing_configuration.json defaults: one tile, k=7, maxCorners 20000
    t_gen = time.perf_counter()
    mon_t, ref_t = synth.make_pair_torch(S, S, 0.5, 0.25, seed=20260101 + 10 * rank, device=dev)
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t_gen
    ctx = Context(local_rank)
    pair = ResidentPair.from_device_pointers(mon_t.data_ptr(), ref_t.data_ptr(), np.uint16, S, S, ctx=ctx, keepalive=(mon_t, ref_t))

    # One step = one band pair through the whole hot path.  The main thread SUBMITS the pair (km_klt_tile_frame_submit: the
    # call returns when the pair's last kernel and the copy of its frame block are enqueued, so the next pair's dense
    # stages queue right behind them); a worker thread waits for the block and runs the host half (frame block -> pandas
    # DataFrame + radial error / angle columns, numpy as in the reference) while the device already works on the next
    # pair (ctypes releases the GIL inside the library).  Every frame is complete before the closing fence, so K timed
    # steps are K finished pairs.
    from concurrent.futures import ThreadPoolExecutor
    pool = ThreadPoolExecutor(max_workers=1)
    stage_sum = {}
    totals = {"rows": 0, "frames": 0, "n_init": 0}

    def host_half(pend):
        raw = pend.wait()
        spans = pend.stage_ms()
        frame = raw.to_frame()
        return raw, spans, (None if frame is None else pair.score_frame(frame, 0.4))

    def collect(pending):
        """Result of an earlier step: its frame, and - the path's only exchange step - the all-gather of every rank's
        key-point block (device pipeline layout) over RCCL; the gathered blocks stay in HBM."""
        raw, spans, frame = pending.result()
        n_rows = raw.n_rows
        if world > 1:
            _, n_rows = gather_rank_blocks(raw.block, conf.maxCorners, True, device=coll_dev)
        totals["rows"] += n_rows
        totals["frames"] += 1
        totals["n_init"] = int(raw.block[:4].view(np.int32)[1])
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
    ctx.set_profiling(False)
    assert totals["frames"] == a.steps
    n_kp_total = totals["rows"]
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
        stage_ms = {k: v / a.steps for k, v in stage_sum.items()}
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
            "n_init": int(stats.n_init), "n_candidates": int(stats.n_candidates), "select_batches": int(stats.n_select_batches),
            "median_dx_dy": (None if frame is None else [float(np.median(frame["dx"])), float(np.median(frame["dy"]))]),
            "stage_ms": {k: round(v, 4) for k, v in stage_ms.items()},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": algo_bytes, "kernel_ms": dense[dom],
                         "dense_path": {"bytes": dense_bytes, "ms": dense_ms, "achieved": dense_bytes / (dense_ms * 1e-3) / 1e9,
                                        "frac": dense_bytes / (dense_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}},
            "synth_seconds": round(t_gen, 2),
        }
    if rank == 0 and world == 1 and not a.no_cpu_baseline:      # reported baseline: rank 0 at N=1 only
        cb, _ = cpu_baseline(mon_t, ref_t, S, dict(maxCorners=conf.maxCorners), a.cpu_sample_rows)
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
