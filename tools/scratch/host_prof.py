import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from karios_amd import synth
from karios_amd._lib import Context
from karios_amd.core import KLTConfiguration
from karios_amd.resident import ResidentPair
from concurrent.futures import ThreadPoolExecutor
S = 10980
dev = torch.device("cuda", 0)
conf = KLTConfiguration()
sys.setswitchinterval(1e-4)
m, r = synth.make_pair_torch(S, S, 0.5, 0.25, device=dev); torch.cuda.synchronize()
ctx = Context(0)
pair = ResidentPair.from_device_pointers(m.data_ptr(), r.data_ptr(), np.uint16, S, S, ctx=ctx, keepalive=(m, r))
pool = ThreadPoolExecutor(max_workers=1)
T = {"wait": 0.0, "stage": 0.0, "frame": 0.0, "score": 0.0, "submit": 0.0, "collect": 0.0}
def host_half(pend):
    a = time.perf_counter(); raw = pend.wait(); b = time.perf_counter(); sp = pend.stage_ms(); c = time.perf_counter()
    f = raw.to_frame(radial=True); d = time.perf_counter(); f = pair.score_frame(f, 0.4); e = time.perf_counter()
    T["wait"] += b - a; T["stage"] += c - b; T["frame"] += d - c; T["score"] += e - d
    return f
for mode in ("threaded", "submit-only"):
    for k in T: T[k] = 0.0
    pending = None
    N = 60
    for i in range(6):
        pair.submit_tile(conf, zncc_threshold=0.4).wait()
    ctx.sync(); t0 = time.perf_counter()
    if mode == "threaded":
        for i in range(N):
            a = time.perf_counter(); p = pair.submit_tile(conf, zncc_threshold=0.4); T["submit"] += time.perf_counter() - a
            nxt = pool.submit(host_half, p)
            a = time.perf_counter()
            if pending is not None: pending.result()
            T["collect"] += time.perf_counter() - a
            pending = nxt
        pending.result()
    else:
        q = []
        for i in range(N):
            a = time.perf_counter(); q.append(pair.submit_tile(conf, zncc_threshold=0.4)); T["submit"] += time.perf_counter() - a
            if len(q) > 2: q.pop(0).wait()
        for p in q: p.wait()
    ctx.sync(); dt = time.perf_counter() - t0
    print(mode, f"{dt/N*1e3:.3f} ms per pair;", {k: round(v / N * 1e3, 3) for k, v in T.items()}, flush=True)
print("cpu count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
