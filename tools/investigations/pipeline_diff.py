#!/usr/bin/env python3
"""Diagnosis: which columns of which unit differ between a pipelined and an unpipelined stream of batched submissions."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from karios_amd import synth, _lib
from karios_amd.core import KLTConfiguration
from karios_amd.resident import ResidentPair, submit_units
ctx = _lib.default_context()
mon_a, ref_a = synth.make_pair(1400, 1500, 0.5, 0.25, seed=11, nodata_wedge=True)
mon_b, ref_b = synth.make_pair(900, 1100, -0.3, 0.4, seed=12)
pa, pb = ResidentPair.upload(mon_a, ref_a, ctx=ctx), ResidentPair.upload(mon_b, ref_b, ctx=ctx)
UNITS_A = [(0, 0, 700, 600), (700, 0, 800, 600), (0, 600, 1500, 800), (300, 200, 640, 512), None]
conf = KLTConfiguration(maxCorners=1200)
bx = [(pa, b, None) for b in UNITS_A[:3]] + [(pb, None, None)]
by = [(pb, (100, 50, 900, 700), (5100, 7050)), (pa, UNITS_A[3], None), (pa, None, None)]
mi = os.environ.get("MI", "1") == "1"
want = {}
for name, units in (("x", bx), ("y", by)):
    b = submit_units(units, conf, 0.4, mi)
    want[name] = b.wait()
names = ["x0", "y0", "dx", "dy", "score", "index"]
def diff(batch, name, tag):
    for k, (g, w) in enumerate(zip(batch.wait(), want[name])):
        ia, ib = g.block.view(np.int32), w.block.view(np.int32)
        n, cap = int(ia[0]), g.cap
        msg = []
        if not np.array_equal(ia[:4], ib[:4]): msg.append(f"hdr {ia[:4].tolist()} {ib[:4].tolist()}")
        for c in range(6):
            a_, b_ = ia[4 + c * cap:4 + c * cap + n], ib[4 + c * cap:4 + c * cap + n]
            if not np.array_equal(a_, b_): msg.append(f"{names[c]}: {int((a_ != b_).sum())} of {n} rows, first at {int(np.nonzero(a_ != b_)[0][0])}")
        base = 4 + 6 * cap
        for c in range(int(g.with_zncc)):
            a_, b_ = ia[base + 2 * c * cap:base + 2 * c * cap + 2 * n], ib[base + 2 * c * cap:base + 2 * c * cap + 2 * n]
            if not np.array_equal(a_, b_): msg.append(f"score column {c}: {int((a_ != b_).sum())} words differ")
        if msg: print(tag, name, "unit", k, "; ".join(msg), flush=True)
ctx.set_option("units_pipeline", int(os.environ.get("PIPE", "1")))
seq = (os.environ.get("SEQ") or "xyxxyyx" * 3)
prev = None
for i, name in enumerate(seq):
    cur = (submit_units(bx if name == "x" else by, conf, 0.4, mi), name, f"step {i}")
    if prev is not None: diff(*prev)
    prev = cur
diff(*prev)
ctx.set_option("units_pipeline", 0)
print("done")
