#!/usr/bin/env python3
"""N contexts in flight, each a FrameStream of batched 4-pair submissions on its own thread: ms per pair over all of them.  Measures what a
second submission in flight buys TODAY.  python tools/two_contexts_probe.py [contexts ...]"""
import os
import sys
import threading
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
REPS = 4
dev = torch.device("cuda", 0)
mon, ref = synth.make_pair_torch(S, S, 0.5, 0.25, device=dev)
torch.cuda.synchronize()
conf = KLTConfiguration()
for n_ctx in [int(v) for v in sys.argv[1:]] or [1, 2]:
    ctxs = [Context(0) for _ in range(n_ctx)]
    if os.environ.get("KARIOS_PROBE_HP") == "1":
        for c in ctxs:
            c.set_option("chain_hp", 1)
    pairs = [ResidentPair.from_device_pointers(mon.data_ptr(), ref.data_ptr(), np.uint16, S, S, ctx=c, keepalive=(mon, ref)) for c in ctxs]
    SUBS = int(os.environ.get("KARIOS_PROBE_SUBS", "50"))
    barrier = threading.Barrier(n_ctx + 1)
    rows = [0] * n_ctx

    def worker(k):
        with FrameStream(0.4, depth=2) as s:
            def go(n):
                r = 0
                for _ in range(n):
                    r += sum(d.raw.n_rows for d in s.submit_many([(pairs[k], None, None)] * 4, conf))
                r += sum(d.raw.n_rows for d in s.drain())
                ctxs[k].sync()
                return r
            go(3)
            for _rep in range(REPS):
                barrier.wait()
                rows[k] = go(SUBS)
                barrier.wait()

    th = [threading.Thread(target=worker, args=(k,)) for k in range(n_ctx)]
    for t in th:
        t.start()
    res = []
    for _rep in range(REPS):
        barrier.wait()
        t0 = time.perf_counter()
        barrier.wait()
        res.append((time.perf_counter() - t0) / (SUBS * 4 * n_ctx) * 1e3)
    for t in th:
        t.join()
    print(f"{n_ctx} context(s), 4 pairs per submission each: {min(res):.4f} ms per pair (runs: {' '.join(f'{v:.4f}' for v in res)}), rows per pair {sum(rows) // (SUBS * 4 * n_ctx)}", flush=True)
    for c in ctxs:
        c.close()
