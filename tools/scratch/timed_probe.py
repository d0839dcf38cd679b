#!/usr/bin/env python3
"""Dev tool: spread of 20-step regions vs FrameStream depth."""
import os, sys, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from karios_amd import synth
from karios_amd._lib import Context
from karios_amd.core import KLTConfiguration
from karios_amd.resident import ResidentPair
from karios_amd.stream import FrameStream
S = 10980
dev = torch.device("cuda", 0)
mon_t, ref_t = synth.make_pair_torch(S, S, 0.5, 0.25, device=dev)
torch.cuda.synchronize()
ctx = Context(0)
pair = ResidentPair.from_device_pointers(mon_t.data_ptr(), ref_t.data_ptr(), np.uint16, S, S, ctx=ctx)
conf = KLTConfiguration()

def region(stream, n):
    ctx.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        stream.submit(pair, conf)
    stream.drain()
    ctx.sync(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

for rep in range(2):
    for depth in (1, int(os.environ.get("DEPTH2", "2"))):
        with FrameStream(0.4, depth=depth, want_spans=True) as st:
            region(st, 20)
            ts = sorted(region(st, 20) for _ in range(25))
            print(f"depth={depth}: 25 regions of 20 steps: min {ts[0]:.4f} median {ts[12]:.4f} p90 {ts[22]:.4f} max {ts[-1]:.4f}")
