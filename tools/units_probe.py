#!/usr/bin/env python3
"""Stage spans of ONE batched submission (km_klt_units_frame_submit) against the same units submitted one by one: the config-4 shape
(4 bands x tile_size 5490 = 16 units) or the e2e shape (one pair, tile_size 6000: four unequal tiles).
    python tools/units_probe.py [config4|e2e|w=<width>] [reps]      (w=<width>: the config-4 shape with 2 x 2 boxes of that side per band)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from karios_amd import synth, tiling
from karios_amd._lib import Context
from karios_amd.core import KLTConfiguration
from karios_amd.resident import ResidentPair, submit_units

shape = sys.argv[1] if len(sys.argv) > 1 else "config4"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
S = 10980
dev = torch.device("cuda", 0)
ctx = Context(0)
side = int(shape[2:]) if shape.startswith("w=") else 0
if shape == "config4" or side:
    conf = KLTConfiguration(tile_size=5490)
    bands = 4
else:
    conf = KLTConfiguration(tile_size=6000, laplacian_kernel_size=5)
    bands = 1
units, keep = [], []
for b in range(bands):
    mon, ref = synth.make_pair_torch(S, S, 0.5, 0.25, seed=20260101 + 10 * b, device=dev)
    torch.cuda.synchronize()
    pair = ResidentPair.from_device_pointers(mon.data_ptr(), ref.data_ptr(), np.uint16, S, S, ctx=ctx, keepalive=(mon, ref))
    keep.append(pair)
    units += ([(pair, (x, y, side, side), None) for x in (0, min(side, S - side)) for y in (0, min(side, S - side))] if side
              else [(pair, tuple(t), None) for t in tiling.tile_grid(S, S, conf.tile_size)])
print(f"{shape}: {len(units)} units", flush=True)


def run_batched(n):
    t0 = time.perf_counter()
    pend = [submit_units(units, conf, 0.4) for _ in range(n)]
    rows = sum(sum(r.n_rows for r in p.wait()) for p in pend) if n <= 3 else 0
    if n > 3:
        for p in pend:
            p.wait()
    ctx.sync()
    return (time.perf_counter() - t0) / n, rows


for _ in range(3):
    submit_units(units, conf, 0.4).wait()
ctx.sync()
ms = []
for _ in range(reps):
    t0 = time.perf_counter()
    a = submit_units(units, conf, 0.4)
    b = submit_units(units, conf, 0.4)
    a.wait(); b.wait()
    ctx.sync()
    ms.append((time.perf_counter() - t0) / 2 * 1e3)
print("batched: ms per batch (two in flight)", [round(v, 3) for v in ms], flush=True)
ctx.set_profiling(True)
ctx.set_option("profile_stage", -1)
ctx.set_option("profile_every", 1)
acc = {}
for _ in range(reps):
    p = submit_units(units, conf, 0.4)
    p.wait()
    for k, v in p.stage_ms().items():
        acc[k] = acc.get(k, 0.0) + v / reps
print("batched stage spans (ms):", {k: round(v, 4) for k, v in acc.items() if v > 0}, flush=True)
acc = {}
for _ in range(2):
    for pair, box, _o in units:
        p = pair.submit_tile(conf, box=box, zncc_threshold=0.4)
        p.wait()
        for k, v in p.stage_ms().items():
            acc[k] = acc.get(k, 0.0) + v / 2
print("one by one, spans summed over the units (ms):", {k: round(v, 4) for k, v in acc.items() if v > 0}, flush=True)
ctx.set_profiling(False)
