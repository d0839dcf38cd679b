"""BASELINE config 4: 4 bands x tile_size 5490 = 16 units split over the ranks, one all-gather per step (strong scaling)."""
from __future__ import annotations

import json
import os
import statistics
import time

import numpy as np

# ---------------------------------------------------------------------------------------------------- config 4
def config4(dev, rank, world, coll_dev, steps, n_ctx_max=3, batched=False, warmup=2, timed_stage=None):
    """4 bands x tile_size 5490 = 16 work units of 10980^2 pairs (seeds 20260101 + 10 b), split round-robin over the ranks;
    every rank keeps only its units' regions (box + ZNCC halo) resident; a step = all 16 units + ONE all-gather of their blocks.
    `batched` (the object's value since round 5): a rank's units go through ONE batched submission (km_klt_units_frame_submit: one set of
    device launches for all of them, blocks straight into the send buffer at its row pitch); else unit by unit on `n_ctx_max` library
    contexts (units in flight fill each other's latency-bound stretches): the A/B in `contexts_in_flight_ab`.
    `warmup` untimed steps, then EXACTLY `steps` timed ones between fences (barrier + synchronize), max over ranks.  `timed_stage`
    (batched mode): that stage of every second submission is bracketed by HIP events inside the timed region -> `stage_launch_ms`."""
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
    send2 = [send, send.clone(), send.clone()] if batched else None      # (three steps may be in flight: two submitted ahead of the one collected)

    def submit_step(k):
        buf = send2[k % 3]
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

    span = {"ms": 0.0, "n": 0, "units": 0, "cand": 0}

    def collect_step(k, batches):
        buf = send2[k % 3]
        for lo, batch in batches:
            raws = batch.wait()
            if timed_stage is not None:
                t = batch.stage_ms().get(timed_stage, 0.0)
                if t > 0:
                    span["ms"] += t
                    span["n"] += 1
                    span["units"] += len(raws)
                    span["cand"] += sum(r.n_candidates for r in raws)
            for i, raw in enumerate(raws):                   # (the blocks have left the device: the sink's copy is ahead of the host slot's)
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
        # steps k + 1 and k + 2 are submitted before step k is collected: the host's collect (wait, gather, header read-back) and the next
        # submission's min / max must not sit between two steps' dense kernels (0.66 ms per step with one step ahead, timeline_r06_c4.txt)
        from collections import deque
        pend, r = deque(), (0, 0)
        trace = [] if os.environ.get("KARIOS_C4_TRACE") == "1" else None      # host-side times of every submit / collect (tools/config4_probe.py)
        for k in range(n):
            t_a = time.perf_counter()
            pend.append((k, submit_step(k) if resident else []))
            t_b = time.perf_counter()
            while len(pend) > 2:
                r = collect_step(*pend.popleft())
            if trace is not None:
                trace.append((k, t_a, t_b, time.perf_counter()))
        while pend:
            r = collect_step(*pend.popleft())
        if trace:
            t0_ = trace[0][1]
            print("config4 host trace (ms): step, submit at, submit took, collect took", flush=True)
            for k, t_a, t_b, t_c in trace:
                print(f"   {k:3d} {1e3 * (t_a - t0_):9.3f} {1e3 * (t_b - t_a):7.3f} {1e3 * (t_c - t_b):7.3f}", flush=True)
        return r

    if batched:
        for c in ctxs:
            c.set_option("units_pipeline", 1)      # consecutive steps interleave on the device (csrc/api_units.hip); collect_step waits for step k - 1 behind step k's submission
    rows, got = run(max(1, warmup))
    if timed_stage is not None and batched:
        names = [ctxs[0].lib.km_stage_name(i).decode() for i in range(16)]
        ctxs[0].set_option("profile_stage", names.index(timed_stage))
        ctxs[0].set_option("profile_every", 2)
        ctxs[0].set_profiling(True)
        run(2)
        span.update(ms=0.0, n=0, units=0, cand=0)
    fence()
    t0 = time.perf_counter()
    rows, got = run(steps)
    fence()
    dt = time.perf_counter() - t0
    if timed_stage is not None and batched:
        ctxs[0].set_profiling(False)
    if world > 1:
        t = torch.tensor([dt], device=coll_dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    n_mine, px_mine = len(resident), float(sum(b[2] * b[3] for _, _, b in resident))
    del resident
    for c in ctxs:
        c.close()
    extra = {}
    if span["n"]:
        extra = {"timed_stage": timed_stage, "stage_launch_ms": span["ms"] / span["n"], "stage_launches_timed": span["n"],
                 "units_per_launch": span["units"] / span["n"], "candidates_per_launch_rank0": span["cand"] / span["n"], "px_per_launch_rank0": px_mine / max(1, -(-n_mine // 16))}
    return {**extra, "workload": "BASELINE config 4: 4 synthetic band pairs 10980x10980 uint16 (seeds 20260101+10b), tile_size 5490 -> 16 units, KLT + ZNCC, "
                        "each rank holds only its units' regions (box + 128 px halo); one all-gather of the 16 frame blocks per step",
            "scaling": "strong", "n_gpus": world, "units": len(units), "units_gathered": got,
            "units_per_rank": [len(units_of_rank(units, r, world)) for r in range(world)], "contexts_in_flight_per_rank": n_ctx,
            "submission": ("batched: one km_klt_units_frame_submit per rank and step, steps pipelined (step k + 1 submitted before step k is collected and gathered)"
                           if batched else f"unit by unit on {n_ctx} context(s), a host synchronisation per step (round 4's loop)"),
            "steps": steps, "ms_per_step": dt / steps * 1e3, "value": 4 * S * S / 1e6 / (dt / steps), "unit": "Mpx/s",
            "matched_keypoints_per_step": rows, "matched_keypoints_per_sec": rows / (dt / steps), "units_repeated_exactly_on_this_rank": redone[0]}
