#!/usr/bin/env python3
"""Dev tool: LK stage time vs number of key points (fill / drain share of the launch)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from karios_amd import ops, synth
from karios_amd.core import KLTConfiguration

S = 4096
mon, ref = synth.make_pair(S, S, 0.5, 0.25)
ctx = ops._lib.default_context()
lr, lm = ops.laplacian_u8(ops.to_uint8(ref), 7), ops.laplacian_u8(ops.to_uint8(mon), 7)
p0 = ops.good_features_to_track(lr, 100000, 0.05, 5, blockSize=15)
print("corners", len(p0))
rng = np.random.default_rng(1)
p0 = p0[rng.permutation(len(p0))]
ctx.set_profiling(True)
for order in (0, 1):
    ctx.set_option("lk_order", order)
    for n in (2500, 5000, 10000, 20000, 40000, 80000):
        if n > len(p0):
            break
        conf = KLTConfiguration(maxCorners=n)
        ts = []
        for _ in range(5):
            ops.klt_track(lr, lm, None, conf, p0=p0[:n])
            ts.append(ctx.stage_ms()["lk_fwd_bwd"])
        print(f"order={order} n={n:6d} lk={min(ts):.4f} ms  per 20000: {min(ts) * 20000 / n:.4f}")
