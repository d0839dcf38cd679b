#!/usr/bin/env python3
"""Dev tool: stage table of the config-5 stand-in (cross-sensor pair + user mask) next to the plain pair."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from karios_amd import synth
from karios_amd._lib import Context
from karios_amd.core import KLTConfiguration
from karios_amd.resident import ResidentPair
from karios_amd.stream import FrameStream
S = 10980
dev = torch.device("cuda", 0)
ctx = Context(0)
conf = KLTConfiguration()
mon_t, ref_t, mask_t = synth.make_cross_sensor_pair_torch(S, S, device=dev)
torch.cuda.synchronize()
for name, mask in (("user mask", mask_t), ("no mask", None)):
    pair = ResidentPair.from_device_pointers(mon_t.data_ptr(), ref_t.data_ptr(), np.uint16, S, S, ctx=ctx, mask_ptr=(mask.data_ptr() if mask is not None else None))
    ctx.set_profiling(True)
    acc = {}
    with FrameStream(0.4, depth=1, want_spans=True) as s:
        for i in range(12):
            for r in s.submit(pair, conf):
                if i >= 4:
                    for k, v in r.spans.items(): acc[k] = acc.get(k, 0) + v
        n = 12 - 4 - 1
        s.drain()
    ctx.set_profiling(False)
    with FrameStream(0.4, depth=1) as s:
        for _ in range(5): s.submit(pair, conf)
        s.drain(); ctx.sync()
        t0 = time.perf_counter()
        for _ in range(30): s.submit(pair, conf)
        s.drain(); ctx.sync()
        dt = (time.perf_counter() - t0) / 30
    print(name, f"{dt*1e3:.4f} ms/pair |", " ".join(f"{k[:8]}={v/n:.3f}" for k, v in acc.items() if v > 0), "| stats", ctx.stats().n_candidates)
