"""The headline loop of the bench: a stream of independent band pairs through the whole hot path, G DISTINCT resident pairs per
batched submission (BASELINE config 2; reference: one band pair per `_compute_matches`, karios/api/core.py:845-871).

One step = one 10980^2 pair: uint8 stretch -> Laplacian(k=7) -> auto mask -> Shi-Tomasi -> pyramidal LK forward / backward -> FB
score -> ZNCC of the rows with score >= 0.4, driven through `karios_amd.stream.FrameStream`.  The G pairs of a submission are G
different rasters (seeds 20260101 + 10 b - the bands config 4 generates), all resident in HBM (G x 482 MB): pair s % G is step s.
"""
from __future__ import annotations

import gc
import os
import time

import numpy as np

from .model import roofline_of


class Env:
    """What the legs share: rank / world, devices, the collective backend, the library context."""

    def __init__(self, **kw):
        self.__dict__.update(kw)


def pair_seed(rank: int, G: int, b: int) -> int:
    return 20260101 + 10 * (rank * G + b)


def make_pairs(env, S: int, G: int):
    """G distinct synthetic pairs of this rank, resident in HBM -> ([(mon_t, ref_t)], [ResidentPair], seconds)."""
    import torch
    from karios_amd import synth
    from karios_amd.resident import ResidentPair
    t0 = time.perf_counter()
    data = [synth.make_pair_torch(S, S, 0.5, 0.25, seed=pair_seed(env.rank, G, b), device=env.dev) for b in range(G)]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    pairs = [ResidentPair.from_device_pointers(m.data_ptr(), r.data_ptr(), np.uint16, S, S, ctx=env.ctx, keepalive=(m, r)) for m, r in data]
    return data, pairs, dt


def pairs_on(ctx, data, S):
    """The same resident rasters as `ResidentPair`s of another library context (a context = streams + workspace; the pixels are shared)."""
    from karios_amd.resident import ResidentPair
    return [ResidentPair.from_device_pointers(m.data_ptr(), r.data_ptr(), np.uint16, S, S, ctx=ctx, keepalive=(m, r)) for m, r in data]


def run(a, env, conf, pairs, S: int, more_pairs=()):
    """W warm-up steps, settle, EXACTLY K timed steps between fences (barrier + synchronize), max over ranks.
    `more_pairs`: the same pairs on further library contexts (`pairs_on`) - consecutive submissions then go to the contexts in turn:
    each context runs its own software pipeline (csrc/api_units.hip) and the GPU interleaves them (N = 1 without the exchange only).
    -> dict of everything measured (rank 0 fills the roofline), including `last_frames` (pair index -> DataFrame of its last step)."""
    import torch
    import torch.distributed as dist
    from karios_amd.parallel import RankBlockExchange
    from karios_amd.stream import FrameStream

    ctx, world, coll_dev = env.ctx, env.world, env.coll_dev
    exchanging = world > 1 or env.force_exchange
    G = max(1, min(int(a.pairs_per_submission), len(pairs)))
    P = len(pairs)
    by_ctx = [pairs] + ([] if exchanging else [list(p) for p in more_pairs])
    ctxs = [p[0].ctx for p in by_ctx]
    n_ctx = len(by_ctx)
    depth = max(0, min(2, a.depth)) * n_ctx         # (a context holds at most three frames in flight)
    sub_no = [0]
    stage_sum = {}
    totals = {"rows": 0, "frames": 0, "redone": 0, "redone_rows": 0, "last": {}, "n_init": {}, "n_cand": {}}
    # the path's only exchange step: one all-gather of every rank's key-point block per step.  RCCL: issued on a side
    # stream behind a DEVICE-side wait for the block, counted on the device, read once behind the last step.
    ex = RankBlockExchange(ctx, conf.maxCorners, True, device=coll_dev, halves=depth + 1) if exchanging else None
    step_no = [0, 0]                    # units submitted / (gloo) units handed to the exchange
    pend_of = {}                        # RCCL exchange: first step of a submission -> (its pending frame / batch, units)
    group_size = {}                     # first step of a submission -> pairs it carries (its stage spans cover all of them)

    def submit_group(n):
        """`n` consecutive steps as ONE submission: n = 1 the single-unit entry point, else a batched submission of n DISTINCT pairs."""
        on_gpu = ex is not None and ex.on_gpu
        k = step_no[0]
        step_no[0] += n
        group_size[k] = n
        if n == 1:
            if on_gpu:
                ex.arm(k)
                return stream.submit(pairs[k % P], conf, tag=k, on_submitted=lambda pend, k=k: pend_of.__setitem__(k, (pend, 1)))
            return stream.submit(pairs[k % P], conf, tag=k)
        if on_gpu:
            ex.arm_many(k, n)
        mine = by_ctx[sub_no[0] % n_ctx]
        sub_no[0] += 1
        return stream.submit_many([(mine[(k + i) % P], None, None) for i in range(n)], conf, tags=list(range(k, k + n)),
                                  on_submitted=(lambda pend, _i, k=k, n=n: pend_of.__setitem__(k, (pend, n))) if on_gpu else None)

    def run_steps(n, marks=None):
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
        """Finished steps: their frames, and the exchange of their blocks - issued when the step has been collected (its block reached
        the send slot in HBM long ago: the side stream's device-side wait is satisfied at once; the host never waits for a collective)."""
        for d in results:
            n_rows = d.raw.n_rows
            if ex is not None and ex.on_gpu:
                if d.tag in pend_of:
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
            b = d.tag % P
            totals["n_init"][b] = int(d.raw.block[:4].view(np.int32)[1])
            totals["n_cand"][b] = d.raw.n_candidates
            totals["redone"] += int(d.redone)
            totals["last"][b] = d.frame
            gs = group_size.pop(d.tag, 1)
            for k, v in d.spans.items():
                stage_sum[k] = stage_sum.get(k, 0.0) + v
            if any(v > 0 for v in d.spans.values()):
                totals["span_samples"] = totals.get("span_samples", 0) + 1
                totals["span_units"] = totals.get("span_units", 0) + gs

    def fence():
        for c in ctxs:
            c.sync()
        torch.cuda.synchronize()
        if exchanging:
            dist.barrier()
            torch.cuda.synchronize()

    def all_agree(flag_value, cap_hit):
        if not exchanging:
            return flag_value
        flag = torch.tensor([1 if flag_value else 0], device=coll_dev, dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return bool(flag.item()) or cap_hit

    class _All:                                   # (options / profiling switches go to every context of the loop)
        def set_option(self, name, value):
            for c in ctxs:
                c.set_option(name, value)

        def set_profiling(self, on):
            for c in ctxs:
                c.set_profiling(on)

    every = _All()
    stream = FrameStream(0.4, depth=depth, want_spans=True)
    run_steps(a.warmup)
    take(stream.drain())
    # settle (untimed, on top of the W warm-up steps): a fresh box ramps its clocks over the first few hundred milliseconds of load -
    # windows of 20 steps until two consecutive ones agree within 2 % (at least 1 s, at most 3 s of work)
    settle = {"windows": 0}
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
        if all_agree(done, settle["windows"] >= 150):
            break
        prev = cur
    settle["seconds"] = round(time.perf_counter() - t_settle, 3)
    settle["last_window_ms_per_step"] = round(cur / 20 * 1e3, 4)
    # HIP events on the library stream bracket ONE stage inside the timed region - the one the roofline is quoted on (a timed span is
    # two event records = two points where consecutive kernels may not overlap); the full stage table comes from an untimed pass after.
    stage_names = [ctx.lib.km_stage_name(i).decode() for i in range(16)]
    # the interpreter's cyclic collector stays out of the timed steps (a generation-2 pass over torch + pandas takes 10 - 20 ms); it is
    # run HERE, in front of the probe steps and a second settle, which bring clocks and caches back under load
    gc.collect()
    gc.disable()
    every.set_option("profile_stage", -1)
    every.set_profiling(True)
    stage_sum.clear()
    run_steps(2 * G)
    take(stream.drain())
    fence()
    probe = {k: v for k, v in stage_sum.items() if k in ("stretch_laplacian_mask", "min_eigen", "lk_fwd_bwd") and v > 0}
    timed_stage = max(probe, key=probe.get) if probe else "min_eigen"
    if a.timed_stage not in ("auto", "none"):
        timed_stage = a.timed_stage
    every.set_option("profile_stage", stage_names.index(timed_stage))
    every.set_option("profile_every", 2)            # the timed steps are SAMPLED: every second submission records the stage's two events
    if a.timed_stage == "none":
        every.set_profiling(False)
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
        if all_agree(done2, settle["post_gc_windows"] >= 60):
            break
        prev2 = cur2
    settle["post_gc_seconds"] = round(time.perf_counter() - t_s2, 3)
    stage_sum.clear()
    totals.update(rows=0, frames=0, redone=0, redone_rows=0, span_samples=0, span_units=0)
    if ex is not None:
        ex.finish()
        ex.reset_counts()
    fence()
    cpu0 = (time.thread_time(), stream.worker_cpu_s, time.process_time())
    t0 = time.perf_counter()
    marks = [(t0, 0)]
    run_steps(a.steps, marks)
    take(stream.drain())              # the last pair's frame: part of the timed region
    exchange = None
    if ex is not None:
        rows_gathered, flagged_blocks = ex.finish()
        redone_rows = 0
        if flagged_blocks:               # (every rank read the same gathered headers: all of them enter the collective, or none)
            extra = torch.tensor([totals["redone_rows"]], device=coll_dev, dtype=torch.int64)
            dist.all_reduce(extra)
            redone_rows = int(extra.item())
        exchange = {"backend": env.backend if world > 1 else "nccl (one-rank group, KARIOS_BENCH_EXCHANGE=1)",
                    "blocks_in": "HBM (km_set_frame_sink -> send ring)" if ex.on_gpu else "host (gloo development run)",
                    "steps_per_collective": ex.batch, "host_waits_per_step": 0 if ex.on_gpu else "lagged (gloo)", "send_ring_slots": ex.slots,
                    "rows_from_gathered_blocks": rows_gathered, "flagged_blocks_gathered": flagged_blocks,
                    "rows_of_exactly_repeated_units": redone_rows, "steps_exchanged": a.steps}
    fence()
    dt = time.perf_counter() - t0
    host_cpu = {"submit_thread_ms_per_step": round((time.thread_time() - cpu0[0]) / a.steps * 1e3, 4),
                "worker_thread_ms_per_step": round((stream.worker_cpu_s - cpu0[1]) / a.steps * 1e3, 4),
                "process_ms_per_step": round((time.process_time() - cpu0[2]) / a.steps * 1e3, 4)}
    gc.enable()
    gaps_in_order = [round(1e3 * (b[0] - a_[0]) / max(1, b[1]), 3) for a_, b in zip(marks, marks[1:])]
    gaps = sorted(gaps_in_order)
    step_spread = {"median_ms": round(gaps[len(gaps) // 2], 4), "max_ms": round(gaps[-1], 4),
                   "drain_ms": round(1e3 * (dt - (marks[-1][0] - t0)), 4), "in_order_ms": gaps_in_order,
                   "note": f"host-side intervals between consecutive submissions inside the timed region, per PAIR ({G} pair(s) per submission)"}
    assert totals["frames"] == a.steps
    n_kp_total = totals["rows"] if exchange is None else exchange["rows_from_gathered_blocks"] + exchange["rows_of_exactly_repeated_units"]
    timed_samples = totals.get("span_samples", 0)
    timed_ms = stage_sum.get(timed_stage, 0.0) / max(1, totals.get("span_units", 0))       # per PAIR (a launch serves the pairs of its submission)
    timed_launch_ms = stage_sum.get(timed_stage, 0.0) / max(1, timed_samples)
    redone_timed = totals["redone"]
    last_frames = dict(totals["last"])
    # untimed pass: every stage bracketed
    every.set_profiling(True)
    every.set_option("profile_stage", -1)
    every.set_option("profile_every", 1)
    stage_steps = max(G, min(a.steps, 12) // G * G)
    stage_sum.clear()
    run_steps(stage_steps)
    take(stream.drain())
    fence()
    stage_piped = {k: v / stage_steps for k, v in stage_sum.items()}
    # ... and the same pass with the software pipeline OFF: every kernel with the GPU to itself (plus the second stream's pyramids /
    # min-max, as in rounds 1 - 5) - what a kernel costs, against what it takes while it shares the GPU with the other lane's chains
    piped_before = ctx.get_option("units_pipeline", 0)
    every.set_option("units_pipeline", 0)
    stage_sum.clear()
    run_steps(stage_steps)
    take(stream.drain())
    fence()
    every.set_option("units_pipeline", piped_before)
    every.set_profiling(False)
    stream.close()
    if ex is not None:
        ex.finish()                    # (the untimed pass armed the frame sink again: gathered, sink off)
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
    host_cpu["cpus_busy_all_ranks"] = round(sum(host_cpu_ranks) / ms_per_step, 3)
    stats = ctx.stats()
    mm_early = bool(int(stats.path_flags) & 32)      # KM_PATH_MM_EARLY: the last unit's min / max ran beside its predecessor's LK

    out = {"value": mpx_per_s, "ms_per_step": ms_per_step, "seconds": dt, "pairs_per_submission": G, "distinct_pairs_resident": P, "depth": depth,
           "matched_keypoints_per_sec": n_kp_total / dt, "exchange": exchange, "host_cpu_ms_per_step": host_cpu,
           "speculative_tiles_redone": int(redone_timed), "settle": settle, "step_spread": step_spread, "last_frames": last_frames,
           "python_gc": "disabled during the timed steps"}
    if env.rank == 0:
        frames = [f for f in last_frames.values() if f is not None]
        n_init = int(round(np.mean(list(totals["n_init"].values())))) if totals["n_init"] else int(stats.n_init)
        n_cand = int(round(np.mean(list(totals["n_cand"].values())))) if totals["n_cand"] else int(stats.n_candidates)
        n_zncc = int(round(np.mean([(f["score"].to_numpy() >= 0.4).sum() for f in frames]))) if frames else 0
        stage_alone = {k: v / stage_steps for k, v in stage_sum.items()}
        stage_ms = dict(stage_piped)
        if timed_ms > 0:
            stage_ms[timed_stage] = timed_ms      # the roofline kernel: its average over the TIMED region
        roof = roofline_of(stage_ms, S, n_init, n_cand, n_zncc, "min_eigen_candidates_fused" if timed_stage == "min_eigen" else timed_stage,
                           minmax_early=mm_early)
        alone = roofline_of(stage_alone, S, n_init, n_cand, n_zncc, None, minmax_early=mm_early)
        if piped_before and roof["kernel"] in alone["kernels"]:
            k_alone = alone["kernels"][roof["kernel"]]
            roof["kernel_ms_alone"], roof["frac_alone"] = k_alone["ms"], k_alone["frac"]
            roof["note"] = ("frac / achieved / kernel_ms: the kernel's launches inside the timed region, where the software pipeline runs the other "
                            "lane's latency-bound chains beside it; *_alone: the same launch with the pipeline off")
        # the whole step against the HBM roof: every stage's algorithmic bytes / the step's wall time
        b_fused = alone["all_stages"]["bytes"]
        fused_eig = alone["kernels"].get("min_eigen_candidates_fused")
        # SURVEY 8(d) prices the eigenvalue map round trip the fusion removed (P3 + P4 = 10 B/px instead of 2 B/px + 8 B per key)
        b_8d = b_fused if fused_eig is None else b_fused - fused_eig["bytes"] + fused_eig["unfused_model"]["bytes"]
        roof["whole_step"] = {"bytes_per_pair": b_fused, "bytes_per_pair_8d_model": b_8d, "ms_per_pair": ms_per_step,
                              "achieved": b_fused / (ms_per_step * 1e-3) / 1e9, "peak": roof["peak"],
                              "frac": b_fused / (ms_per_step * 1e-3) / 1e9 / roof["peak"],
                              "frac_8d_model": b_8d / (ms_per_step * 1e-3) / 1e9 / roof["peak"],
                              "note": "every stage's must-move bytes / the step's wall time; 8d model: SURVEY 8(d)'s 23.5 B/px + sparse terms"}
        if G > 1:
            # a launch of the batched pipeline serves the G pairs of its submission: bytes and duration both scale by G; `achieved` is
            # bytes per launch / launch duration either way; `kernel_ms` (and the stage table) are quoted per PAIR
            roof["pairs_per_launch"] = G
            roof["launch_ms"] = round(timed_launch_ms, 4) if timed_launch_ms > 0 else round(roof["kernel_ms"] * G, 4)
            roof["algorithmic_bytes_per_launch"] = roof["algorithmic_bytes_per_launch"] * G
            if roof.get("traffic"):
                roof["traffic"] = roof["traffic"] * G
        out.update({
            "matched_keypoints_per_pair": int(round(np.mean([len(f) for f in frames]))) if frames else 0,
            "matched_keypoints_by_pair": {str(b): (0 if f is None else len(f)) for b, f in sorted(last_frames.items())},
            "n_init": n_init, "n_candidates": n_cand,
            "median_dx_dy": (None if not frames else [float(np.median(frames[0]["dx"])), float(np.median(frames[0]["dy"]))]),
            "stage_ms": {k: round(v, 4) for k, v in stage_ms.items()},
            "stage_ms_alone": {k: round(v, 4) for k, v in stage_alone.items()},
            "stage_ms_note": f"per PAIR.  stage_ms: the pipelined loop ({timed_stage}: HIP events on {timed_samples} submissions of the timed region, every "
                             f"second one; the other stages: an untimed pass of {stage_steps} steps right after - a stage's span includes what the other "
                             "lane's chains take from it); stage_ms_alone: the same pass with the software pipeline off (bracketing every stage costs "
                             "~0.07 ms per pair)",
            "roofline": roof})
    return out
