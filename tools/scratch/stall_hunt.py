#!/usr/bin/env python3
"""Dev tool: long headline loop; reports every step whose host-side interval exceeds 2.5 ms, with the unit's redone flag."""
import os, sys, time, gc
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from karios_amd import synth
from karios_amd._lib import Context
from karios_amd.core import KLTConfiguration
from karios_amd.resident import ResidentPair
from karios_amd.stream import FrameStream
S = 10980
N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda", 0)
mon_t, ref_t = synth.make_pair_torch(S, S, 0.5, 0.25, device=dev)
torch.cuda.synchronize()
ctx = Context(0)
pair = ResidentPair.from_device_pointers(mon_t.data_ptr(), ref_t.data_ptr(), np.uint16, S, S, ctx=ctx)
conf = KLTConfiguration()
stream = FrameStream(0.4, depth=depth)
for _ in range(50): stream.submit(pair, conf)
stream.drain()
if len(sys.argv) > 3 and sys.argv[3] == "nogc":
    gc.collect(); gc.disable()
t_prev = time.perf_counter(); slow = []; redone = 0; gaps = []
for i in range(N):
    res = stream.submit(pair, conf)
    t = time.perf_counter()
    g = 1e3 * (t - t_prev); t_prev = t
    gaps.append(g)
    r = any(x.redone for x in res)
    redone += r
    if g > 2.5: slow.append((i, round(g, 2), r, [int(x.raw.flags) for x in res]))
stream.drain()
gaps = np.array(gaps)
print(f"steps {N} depth {depth} mean {gaps.mean():.4f} median {np.median(gaps):.4f} p99 {np.percentile(gaps, 99):.3f} max {gaps.max():.2f} redone {redone} units_redone {stream.units_redone}")
print("slow steps (index, ms, redone, flags):", slow[:40])
