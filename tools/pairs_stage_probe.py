#!/usr/bin/env python3
"""Stage spans of a batched submission of N full-size pairs (one set of launches) - python tools/pairs_stage_probe.py [N ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from karios_amd import synth
from karios_amd._lib import Context
from karios_amd.core import KLTConfiguration
from karios_amd.resident import ResidentPair, submit_units

S = 10980
dev = torch.device("cuda", 0)
ctx = Context(0)
conf = KLTConfiguration()
pairs = []
for b in range(4):
    mon, ref = synth.make_pair_torch(S, S, 0.5, 0.25, seed=20260101 + 10 * b, device=dev)
    torch.cuda.synchronize()
    pairs.append(ResidentPair.from_device_pointers(mon.data_ptr(), ref.data_ptr(), np.uint16, S, S, ctx=ctx, keepalive=(mon, ref)))
for n in [int(v) for v in sys.argv[1:]] or [2, 4]:
    units = [(pairs[i % 4], None, None) for i in range(n)]
    for _ in range(3):
        submit_units(units, conf, 0.4).wait()
    ctx.sync()
    w = []
    for _ in range(5):
        t0 = time.perf_counter()
        a = submit_units(units, conf, 0.4); b = submit_units(units, conf, 0.4)
        a.wait(); b.wait(); ctx.sync()
        w.append((time.perf_counter() - t0) / (2 * n) * 1e3)
    ctx.set_profiling(True); ctx.set_option("profile_stage", -1); ctx.set_option("profile_every", 1)
    acc = {}
    for _ in range(5):
        p = submit_units(units, conf, 0.4); p.wait()
        for k, v in p.stage_ms().items():
            acc[k] = acc.get(k, 0.0) + v / 5 / n
    ctx.set_profiling(False)
    print(f"{n} pairs per submission: ms per pair (two batches in flight) {sorted(w)[2]:.4f}; spans per pair {({k: round(v, 4) for k, v in acc.items() if v > 0})}", flush=True)
