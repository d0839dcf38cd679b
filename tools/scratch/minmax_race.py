#!/usr/bin/env python3
"""Dev tool: hunts the rare stale-input mismatch of the blocking host-buffer tile call: random pairs of changing pixel type and size
through ops.klt_tile, the call's own min / max statistics against numpy."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from karios_amd import ops
from karios_amd._lib import default_context
from karios_amd.core import KLTConfiguration
T = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
conf = KLTConfiguration(maxCorners=200)
ctx = default_context()
t0 = time.time(); n = bad = 0
while time.time() - t0 < T:
    dt = [np.uint16, np.int16, np.float32, np.uint16, np.int16][int(rng.integers(0, 5))]
    H, W = int(rng.integers(200, 1500)), int(rng.integers(200, 1500))
    base = rng.integers(1000, 6000, (H, W)).astype(np.float32)
    ref = (base + rng.normal(0, 50, (H, W))).astype(dt)
    mon = (np.roll(base, 1, 1) + rng.normal(0, 50, (H, W)) + float(rng.integers(-500, 500))).astype(dt)
    y0, x0 = int(rng.integers(0, H // 3)), int(rng.integers(0, W // 3))
    rb, mb = ref[y0:, x0:], mon[y0:, x0:]
    for rep in range(2):
        ops.klt_tile(rb, mb, conf, mon_ksize=3, ref_ksize=3)
        st = ctx.stats()
        got = (st.min_ref, st.max_ref, st.min_mon, st.max_mon)
        want = (float(rb.min()), float(rb.max()), float(mb.min()), float(mb.max()))
        n += 1
        if got != want:
            bad += 1
            print(f"MISMATCH call {n} rep {rep} dtype {np.dtype(dt).name} shape {rb.shape} got {got} want {want} flags {st.path_flags}", flush=True)
print(f"minmax_race seed {seed}: {n} calls, {bad} mismatching, {time.time() - t0:.0f} s")
