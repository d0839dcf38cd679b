#!/usr/bin/env python3
"""A stream of independent full-size pairs (bands of a product) submitted N pairs per batched submission against one pair per submission:
ms per PAIR through FrameStream (depth 2).  python tools/pairs_batched_probe.py [pairs per submission ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
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
pairs = []
for b in range(4):
    mon, ref = synth.make_pair_torch(S, S, 0.5, 0.25, seed=20260101 + 10 * b, device=dev)
    torch.cuda.synchronize()
    pairs.append(ResidentPair.from_device_pointers(mon.data_ptr(), ref.data_ptr(), np.uint16, S, S, ctx=ctx, keepalive=(mon, ref)))
for n in [int(v) for v in sys.argv[1:]] or [1, 2, 4]:
    with FrameStream(0.4, depth=2) as s:
        def go(steps):
            rows = 0
            for k in range(steps):
                units = [(pairs[(k * n + i) % 4], None, None) for i in range(n)]
                res = s.submit_many(units, conf) if n > 1 else s.submit(units[0][0], conf)
                rows += sum(d.raw.n_rows for d in res)
            rows += sum(d.raw.n_rows for d in s.drain())
            ctx.sync()
            return rows
        go(8)
        w = []
        for _ in range(3):
            steps = int(os.environ.get("KARIOS_PROBE_SUBS", max(4, 48 // n)))      # submissions per window
            t0 = time.perf_counter()
            rows = go(steps)
            w.append((time.perf_counter() - t0) / (steps * n) * 1e3)
        print(f"{n} pair(s) per submission: ms per pair {sorted(w)[1]:.4f} (windows {[round(v, 4) for v in w]}), rows per pair {rows // (steps * n)}", flush=True)
