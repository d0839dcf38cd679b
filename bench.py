#!/usr/bin/env python3
"""Benchmark of the KARIOS matching hot path on MI355X (BASELINE.json metric).

`python bench.py --gpus N --steps K --warmup W`: the LAST stdout line is ONE small JSON object (< 6 KB, `benchkit/summary.py`) - the
driver's contract keys, `roofline`, `cpu_baseline`, one-number summaries of the side objects and the in-run gates.  Everything else
(stage tables, gates, sensitivity workloads, configs 3 / 4 / 5 in full) is `detail`: written to `bench_detail.json` (under
`gpurun_out/` when that directory exists) and printed on EARLIER stdout lines as `detail <name> <json>`.  With N > 1 and no launcher
environment the process SPAWNS N ranks itself (before anything touches the GPU); under an external launcher
(`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`) WORLD_SIZE must equal --gpus.

N = 1: `value` = BASELINE config 2.  One step = one synthetic Sentinel-2-sized pair resident in HBM through the whole hot path
(uint8 stretch -> Laplacian k=7 -> auto mask -> Shi-Tomasi -> pyramidal LK fwd / bwd -> FB score -> ZNCC of the rows with score >= 0.4;
default processing_configuration.json: one 10980^2 tile, maxCorners 20000) driven through `karios_amd.stream.FrameStream`; FOUR DISTINCT
resident pairs (seeds 20260101 + 10 b) travel per batched submission - reference: one band pair per `_compute_matches`
(karios/api/core.py:845-871), the four 10 m bands of a product.  `benchkit/headline.py`.

N > 1: `value` = BASELINE config 4 as a FIXED workload - 4 bands x tile_size 5490 = 16 units (karios/matcher/klt.py:220-253) dealt
round-robin to the ranks, batched per rank, ONE all-gather of the 16 frame blocks per step: STRONG scaling (`benchkit/config4.py`).
The band-per-rank stream of independent pairs (weak scaling, exchange without the host in the loop) is the side number `weak_pairs`.

Modules: benchkit/{headline,config4,legs,sensitivity,cpu,model,summary,launcher}.py.  `--config 3` prints the config-3 line of its own.
"""
from __future__ import annotations

import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from benchkit import summary  # noqa: E402
from benchkit.launcher import launch_ranks  # noqa: E402
from benchkit.model import SENS_FILE  # noqa: E402

METRIC = "Mpixels/sec (+ matched keypoints/sec), Sentinel-2 10980^2 pair, KLT + ZNCC"
DTYPE = "u8/i32 stencils, f32 LK solve, f64 stretch + ZNCC"


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", type=int, default=2, choices=(2, 3), help="BASELINE config of the line (3: large-shift pre-alignment, N = 1 only)")
    ap.add_argument("--size", type=int, default=10980, help="image side (BASELINE: 10980)")
    ap.add_argument("--share-gpu", action="store_true", help="development: the N ranks share the visible GPU(s) round-robin and the "
                    "collectives run on gloo / host memory (RCCL wants one device per rank)")
    for leg in ("in-flight", "cpu-baseline", "end-to-end", "full-scoring", "config3", "config4", "config5", "one-pair", "sensitivity", "auto-ksize", "weak-pairs"):
        ap.add_argument(f"--no-{leg}", action="store_true")
    ap.add_argument("--headline-only", action="store_true", help="profiler runs: only the headline's kernels")
    ap.add_argument("--cpu-runs", type=int, default=4, help="timed oracle passes: pass k runs (and gates) distinct pair k %% pairs")
    ap.add_argument("--depth", type=int, default=2, help="submissions still pending when submit() returns (FrameStream depth, <= 2 on one context)")
    ap.add_argument("--pairs-per-submission", type=int, default=4, choices=(1, 2, 4),
                    help="DISTINCT band pairs that travel in ONE batched submission (FrameStream.submit_many -> km_klt_units_frame_submit); "
                         "1: one pair per submission (rounds 1 - 4), which the line reports too (`one_pair_per_submission_ms`)")
    ap.add_argument("--contexts", type=int, default=1, choices=(1, 2, 3),
                    help="library contexts (streams + workspace) the N = 1 headline's submissions go to in turn, each with its own software pipeline "
                         "(the rasters are shared).  Default 1: submitted in turn from ONE thread two contexts measure 0.93 ms per pair against 0.87 "
                         "(three: 0.99); a thread per context (tools/two_contexts_probe.py) gains 4 % over one context (0.785 against 0.822)")
    ap.add_argument("--timed-stage", default="auto", help="stage bracketed by HIP events inside the timed region (auto: the largest kernel; none)")
    ap.add_argument("--detail-file", default=None, help="where the full detail goes (default: gpurun_out/bench_detail.json if that directory exists, else ./bench_detail.json)")
    a = ap.parse_args(argv)
    if a.headline_only:
        for leg in ("in_flight", "cpu_baseline", "end_to_end", "full_scoring", "config3", "config4", "config5", "one_pair", "sensitivity", "auto_ksize", "weak_pairs"):
            setattr(a, "no_" + leg, True)
    return a


def emit(detail: dict, a) -> None:
    """Detail to its file and to earlier stdout lines; the small line LAST."""
    path = a.detail_file or os.path.join(ROOT, "gpurun_out" if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else "", "bench_detail.json")
    detail["detail_where"] = f"{os.path.relpath(path, ROOT)}; `detail <name> {{...}}` lines above this one"
    try:
        with open(path, "w") as f:
            json.dump(detail, f)
    except OSError as e:
        detail["detail_where"] = f"`detail <name> {{...}}` lines above this one (file not written: {e})"
    # RCCL writes a banner through C stdio, flushed at exit - behind everything Python printed.  Flush it out now: the JSON line stays LAST.
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()
    small = summary.small_line(detail)
    core = set(small) | {"detail_where"}
    for k, v in detail.items():
        if isinstance(v, (dict, list)) and k not in ("config",):
            print(f"detail {k} {json.dumps(v)}")
    print("detail scalars " + json.dumps({k: v for k, v in detail.items() if not isinstance(v, (dict, list)) and k not in core}))
    print(json.dumps(small), flush=True)


def init_ranks(a):
    """Process group + device of this rank -> benchkit.headline.Env (ctx attached by the caller)."""
    import torch
    import torch.distributed as dist
    from benchkit.headline import Env
    rank, world, local_rank = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        print(f"bench.py: --gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks", file=sys.stderr)
        sys.exit(2)
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
    # KARIOS_BENCH_EXCHANGE=1: the N > 1 code path of the pair stream - process group, RCCL all-gather of the frame blocks, device-side
    # counting - with a group of ONE rank (tests/test_gpu_bench.py, tests/test_gpu_rccl.py)
    force_exchange = world == 1 and os.environ.get("KARIOS_BENCH_EXCHANGE") == "1"
    if world > 1 or force_exchange:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        if backend == "nccl":
            os.environ.setdefault("NCCL_DEBUG", "NONE")          # RCCL's version banner goes to STDOUT otherwise; an explicit NCCL_DEBUG wins
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
    return Env(rank=rank, world=world, dev=dev, dev_index=dev_index, coll_dev=coll_dev, backend=backend, force_exchange=force_exchange,
               ranks_seen=ranks_seen, devices=devices, share=share)


def main():
    a = parse()
    if a.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        sys.exit(launch_ranks(a))
    env = init_ranks(a)
    import torch
    import torch.distributed as dist
    from benchkit import headline
    from benchkit.config4 import config4
    from karios_amd._lib import Context
    from karios_amd.core import KLTConfiguration

    rank, world, dev = env.rank, env.world, env.dev
    env.ctx = ctx = Context(env.dev_index)
    S = a.size
    launcher = os.environ.get("KARIOS_BENCH_LAUNCHER", "external (torch.distributed.run)" if "WORLD_SIZE" in os.environ else "single process")
    common = {"metric": METRIC, "unit": "Mpx/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "higher_is_better": True, "vs_baseline": None,
              "dtype": DTYPE, "data": "synthetic", "world": world, "launcher": launcher,
              "backend": env.backend if (world > 1 or env.force_exchange) else None, "rccl_ranks_seen": env.ranks_seen, "devices": env.devices}

    if a.config == 3:
        if world > 1:
            raise SystemExit("config 3 (a global 2-D FFT) does not shard: replicas only, run it with --gpus 1")
        from benchkit.legs import config3_object
        o = config3_object(ctx, dev, S, a.steps, a.warmup)
        detail = dict(common, metric="Mpixels/sec (+ matched keypoints/sec), Sentinel-2 10980^2 pair, large-shift pre-alignment + KLT",
                      value=o["value"], ms_per_step=o["ms_per_step"], scaling="weak", dtype=o["dtype"],
                      config={"workload": o["workload"], "pairs_per_step": 1}, roofline=o["roofline"], cpu_baseline=None, config3=o)
        emit(detail, a)
        return

    conf = KLTConfiguration()  # processing_configuration.json defaults: one tile, k=7, maxCorners 20000
    G = max(1, int(a.pairs_per_submission))
    solo = rank == 0 and world == 1
    detail = None

    c4_is_value = world > 1 and not a.no_config4 and S == 10980          # (the same on every rank)
    if c4_is_value:
        # ---------------------------------------------------------------- N > 1: value = config 4, strong scaling
        c4 = config4(dev, rank, world, env.coll_dev, a.steps, batched=True, warmup=a.warmup, timed_stage="min_eigen")
        if rank == 0:
            from benchkit.model import roofline_units
            detail = dict(common, value=c4["value"], ms_per_step=c4["ms_per_step"], scaling="strong",
                          config={"workload": "BASELINE config 4: 4 bands 10980^2 u16 x tile_size 5490 = 16 units over the ranks, KLT + ZNCC, one all-gather per step",
                                  "units": 16, "units_per_rank": c4["units_per_rank"], "pairs_per_step": 4,
                                  "parallelism": f"16 independent units round-robin over {world} ranks, batched per rank, flat all-gather of the frame blocks"},
                          units_per_rank=c4["units_per_rank"], matched_keypoints_per_sec=c4["matched_keypoints_per_sec"],
                          matched_keypoints_per_pair=c4["matched_keypoints_per_step"] // 4, roofline=roofline_units(c4), cpu_baseline=None, config4=c4)
        torch.cuda.empty_cache()

    if not (c4_is_value and a.no_weak_pairs):
        # ---------------------------------------------------------------- the stream of independent band pairs (value at N = 1)
        import copy
        data, pairs, t_gen = headline.make_pairs(env, S, G)
        aa = a
        if c4_is_value:                                  # side number: a short run of the weak-scaling stream
            aa = copy.copy(a)
            aa.steps, aa.warmup = max(G, min(a.steps, 12)), min(a.warmup, 4)
        extra_ctxs = [Context(env.dev_index) for _ in range(max(0, a.contexts - 1))] if world == 1 and not env.force_exchange else []
        head = headline.run(aa, env, conf, pairs, S, more_pairs=[headline.pairs_on(c, data, S) for c in extra_ctxs])
        for c in extra_ctxs:
            c.close()
        last_frames = head.pop("last_frames")
        if rank == 0:
            head_cfg = {"workload": f"BASELINE config 2: {G} distinct S2-sized pairs {S}x{S} u16 resident in HBM, one per step, {G} per batched submission, KLT + ZNCC",
                        "pairs_per_step": world, "pairs_per_submission": head["pairs_per_submission"], "distinct_pairs_resident": head["distinct_pairs_resident"],
                        "contexts": 1 + len(extra_ctxs),
                        "seeds": [headline.pair_seed(0, G, b) for b in range(G)], "klt": "Laplacian k=7, maxCorners 20000, one tile, ZNCC of rows with score >= 0.4",
                        "parallelism": f"{world} rank(s), each a stream of independent band pairs" + (", RCCL all-gather of key-point frames" if world > 1 else "")}
            if detail is None:
                detail = dict(common, scaling="weak", config=head_cfg, cpu_baseline=None, **head)
                detail["synth_seconds"] = round(t_gen, 2)
            else:
                detail["weak_pairs"] = dict(head, steps=aa.steps, config=head_cfg, scaling="weak")
        if solo:
            run_solo_legs(a, env, conf, detail, data, pairs, last_frames, S, G)
        del pairs, data
        torch.cuda.empty_cache()

    if world == 1 and not a.no_config4 and S == 10980:
        k4 = max(3, min(8, a.steps // 3))
        c4 = config4(dev, rank, world, env.coll_dev, k4, batched=True)
        # A/B: the rank's units through ONE batched submission against unit by unit on one / three library contexts
        c4_one = config4(dev, rank, world, env.coll_dev, k4, n_ctx_max=1)
        c4_three = config4(dev, rank, world, env.coll_dev, k4, n_ctx_max=3)
        c4["contexts_in_flight_ab"] = {"batched": {"ms_per_step": c4["ms_per_step"]}, "1": {"ms_per_step": c4_one["ms_per_step"]},
                                       "3": {"ms_per_step": c4_three["ms_per_step"]}, "note": "same 16 units, same box, back to back"}
        detail["config4"] = c4
    if solo and not a.no_cpu_baseline:
        from benchkit.cpu import cpu_baseline
        cb = cpu_baseline(env.host_pair, env.n_host_pairs, dict(maxCorners=conf.maxCorners), a.cpu_runs, gpu_frames=env.last_frames,
                          scored_frame=env.scored_frame)
        if "full_scoring" in detail:
            detail["full_scoring"]["parity"] = cb.pop("full_scoring_parity", {"checked": False})
        detail["cpu_baseline"] = cb
        detail["speedup_vs_cpu_port"] = detail["value"] / cb["value"]
    if world > 1 or env.force_exchange:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        sens = os.path.join(ROOT, SENS_FILE)
        if os.path.exists(sens):
            try:
                detail["oracle_sensitivity"] = dict(json.load(open(sens)), source=f"precomputed: {SENS_FILE} (tools/investigations/oracle_sensitivity.py, round 2)")
            except Exception:
                pass
        emit(detail, a)


def run_solo_legs(a, env, conf, detail, data, pairs, last_frames, S, G):
    """Rank 0 at N = 1: the side objects of the line (each its own leg of benchkit/), and what the cpu_baseline leg needs later."""
    import torch
    from benchkit import legs, sensitivity
    ctx, dev = env.ctx, env.dev
    k = a.steps
    env.last_frames, env.scored_frame, env.n_host_pairs = last_frames, None, len(data)
    # host copies for the oracle / the end-to-end leg are fetched while the rasters are still resident (482 MB per pair)
    host = {}
    if not a.no_cpu_baseline:
        for b in range(min(len(data), max(1, a.cpu_runs))):
            host[b] = tuple(t.cpu().numpy().view(np.uint16) for t in data[b])
    elif not a.no_end_to_end:
        host[0] = tuple(t.cpu().numpy().view(np.uint16) for t in data[0])
    env.host_pair = lambda b: host[b]
    if G > 1 and not a.no_one_pair:
        # the loop of rounds 1 - 4: ONE pair per submission (km_klt_tile_frame_submit), and round 5's form: the SAME pair G times per submission
        one = sensitivity._stream_timing(ctx, pairs[0], conf, S, max(8, min(20, k)))
        detail["one_pair_per_submission"] = dict({kk: one[kk] for kk in ("ms_per_pair", "windows_ms_per_pair", "Mpx_per_s", "stage_ms", "units_repeated_exactly")},
                                                 note="FrameStream.submit(pair) per step: every unit pays its corner-selection chain, its LK fill + drain and its frame launches alone")
        same = sensitivity._stream_timing(ctx, pairs[0], conf, S, max(8, min(20, k)), group=G)
        detail["same_pair_repeated"] = dict({kk: same[kk] for kk in ("ms_per_pair", "windows_ms_per_pair")},
                                            note=f"round 5's headline form: ONE resident pair {G} times per batched submission (cache reuse between the units)")
    if not a.no_full_scoring:
        detail["full_scoring"], env.scored_frame = legs.full_scoring(ctx, pairs[0], conf, S, max(6, min(20, k)), pairs=pairs)
    if not a.no_auto_ksize and hasattr(legs, "auto_ksize_object"):
        detail["auto_ksize"] = legs.auto_ksize_object(ctx, pairs[0], data[0], S)
    if not a.no_in_flight:
        detail["in_flight"] = legs.in_flight(dev, conf, S, data)
    if not a.no_end_to_end:
        detail["end_to_end"] = legs.end_to_end(host[0][0], host[0][1], ctx, max(4, min(12, k)))
    del pairs[:], data[:]
    torch.cuda.empty_cache()
    if not a.no_config3:
        detail["config3"] = legs.config3_object(ctx, dev, S, max(3, min(6, k // 3)), 2)
        torch.cuda.empty_cache()
    if not a.no_config5:
        detail["config5"] = legs.config5_object(ctx, dev, S, max(4, min(10, k // 2)))
        torch.cuda.empty_cache()
    if not a.no_sensitivity:
        detail.update(sensitivity.sensitivity_objects(ctx, dev, S, max(8, min(24, k)), group=G))
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
