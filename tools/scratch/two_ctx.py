import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from karios_amd import synth
from karios_amd._lib import Context
from karios_amd.core import KLTConfiguration
from karios_amd.resident import ResidentPair
S = 10980
dev = torch.device("cuda", 0)
conf = KLTConfiguration()
sys.setswitchinterval(1e-4)
for n_ctx in (1, 2, 3):
    data = [synth.make_pair_torch(S, S, 0.5, 0.25, seed=20260101 + 10 * i, device=dev) for i in range(n_ctx)]
    torch.cuda.synchronize()
    ctxs = [Context(0) for _ in range(n_ctx)]
    pairs = [ResidentPair.from_device_pointers(m.data_ptr(), r.data_ptr(), np.uint16, S, S, ctx=c, keepalive=(m, r)) for (m, r), c in zip(data, ctxs)]
    from concurrent.futures import ThreadPoolExecutor
    pool = ThreadPoolExecutor(max_workers=1)
    def host_half(pair, pend):
        raw = pend.wait()
        f = raw.to_frame(radial=True)
        return pair.score_frame(f, 0.4)
    def run(n):
        futs = []
        rows = 0
        for i in range(n):
            p = pairs[i % n_ctx]
            futs.append(pool.submit(host_half, p, p.submit_tile(conf, zncc_threshold=0.4)))
            if len(futs) > 2 * n_ctx:
                rows += len(futs.pop(0).result())
        for f in futs: rows += len(f.result())
        for c in ctxs: c.sync()
        return rows
    run(6 * n_ctx)
    t0 = time.perf_counter(); N = 60
    rows = run(N)
    dt = time.perf_counter() - t0
    print(f"{n_ctx} context(s): {dt / N * 1e3:.3f} ms per pair, rows {rows}", flush=True)
    pool.shutdown(); del pairs, ctxs, data
print("---- eig stage time under overlap (profile_stage = min_eigen) ----")
for n_ctx in (1, 2, 3):
    data = [synth.make_pair_torch(S, S, 0.5, 0.25, seed=20260101 + 10 * i, device=dev) for i in range(n_ctx)]
    torch.cuda.synchronize()
    ctxs = [Context(0) for _ in range(n_ctx)]
    for c in ctxs:
        c.set_option("profile_stage", 2); c.set_profiling(True)
    pairs = [ResidentPair.from_device_pointers(m.data_ptr(), r.data_ptr(), np.uint16, S, S, ctx=c, keepalive=(m, r)) for (m, r), c in zip(data, ctxs)]
    pend = []
    eig = []
    t0 = None
    for i in range(66):
        if i == 6: 
            for c in ctxs: c.sync()
            t0 = time.perf_counter(); eig.clear()
        p = pairs[i % n_ctx]
        pend.append(p.submit_tile(conf, zncc_threshold=0.4))
        if len(pend) > 2 * n_ctx - 1:
            q = pend.pop(0); q.wait(); eig.append(q.stage_ms().get("min_eigen", 0))
    for q in pend: q.wait(); eig.append(q.stage_ms().get("min_eigen", 0))
    for c in ctxs: c.sync()
    dt = time.perf_counter() - t0
    print(f"{n_ctx} context(s): {dt / 60 * 1e3:.3f} ms per pair (no host stage), eig stage {np.mean(eig[-50:]):.4f} ms", flush=True)
    del pairs, ctxs, data
