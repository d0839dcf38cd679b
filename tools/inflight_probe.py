#!/usr/bin/env python3
"""Throughput with several band pairs in flight on ONE GPU (one context + stream + workspace per pair, one host thread
each): the read-back waits of one pair's corner selection are filled by the other pair's kernels.
    python tools/inflight_probe.py [--inflight 1 2 3] [--pairs 40]"""
import argparse
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--inflight", type=int, nargs="+", default=[1, 2, 3])
    ap.add_argument("--pairs", type=int, default=40)
    ap.add_argument("--size", type=int, default=10980)
    a = ap.parse_args()
    import torch
    from karios_amd import synth
    from karios_amd._lib import Context
    from karios_amd.core import KLTConfiguration
    from karios_amd.resident import ResidentPair
    S = a.size
    dev = torch.device("cuda", 0)
    conf = KLTConfiguration()
    n_max = max(a.inflight)
    data = [synth.make_pair_torch(S, S, 0.5, 0.25, seed=20260101 + 10 * i, device=dev) for i in range(n_max)]
    torch.cuda.synchronize()
    ctxs = [Context(0) for _ in range(n_max)]
    pairs = [ResidentPair.from_device_pointers(m.data_ptr(), r.data_ptr(), np.uint16, S, S, ctx=c, keepalive=(m, r))
             for (m, r), c in zip(data, ctxs)]

    def worker(pair, n, out):
        rows = 0
        for _ in range(n):
            raw = pair.match_tile_raw(conf, zncc_threshold=0.4)
            frame = raw.to_frame()
            if frame is not None:
                frame = pair.score_frame(frame, 0.4)
                rows += len(frame)
        out.append(rows)

    for p in pairs:                       # warm-up: workspaces, code objects
        worker(p, 2, [])
    for n_if in a.inflight:
        per = a.pairs // n_if
        out = []
        th = [threading.Thread(target=worker, args=(pairs[i], per, out)) for i in range(n_if)]
        for c in ctxs:
            c.sync()
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        for c in ctxs[:n_if]:
            c.sync()
        dt = time.perf_counter() - t0
        done = per * n_if
        print(f"in flight {n_if}: {done} pairs in {dt*1e3:.1f} ms = {dt/done*1e3:.3f} ms/pair, {S*S/1e6*done/dt:.0f} Mpx/s, "
              f"{sum(out)} rows", flush=True)


if __name__ == "__main__":
    main()
