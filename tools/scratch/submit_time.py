import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from karios_amd import synth
from karios_amd._lib import Context
from karios_amd.core import KLTConfiguration
from karios_amd.resident import ResidentPair
S = 10980
dev = torch.device("cuda", 0)
mon_t, ref_t = synth.make_pair_torch(S, S, 0.5, 0.25, device=dev); torch.cuda.synchronize()
ctx = Context(0)
pair = ResidentPair.from_device_pointers(mon_t.data_ptr(), ref_t.data_ptr(), np.uint16, S, S, ctx=ctx)
conf = KLTConfiguration()
for spec in (1, 0, 1):
    ctx.set_option("speculative", spec)
    for _ in range(3):
        pair.submit_tile(conf, zncc_threshold=0.4).wait()
    ctx.sync()
    N = 30
    t0 = time.perf_counter(); host = 0.0
    pend = None
    for i in range(N):
        a = time.perf_counter()
        p = pair.submit_tile(conf, zncc_threshold=0.4)
        host += time.perf_counter() - a
        if pend is not None: pend.wait()
        pend = p
    pend.wait(); ctx.sync()
    dt = time.perf_counter() - t0
    print(f"speculative={spec}: {1e3*dt/N:.3f} ms per tile, host time inside submit_tile {1e3*host/N:.3f} ms")
print("---- with stage events (set_profiling) ----")
ctx.set_profiling(True)
for spec in (1, 0):
    ctx.set_option("speculative", spec)
    N = 30
    t0 = time.perf_counter()
    pend = None
    for i in range(N):
        p = pair.submit_tile(conf, zncc_threshold=0.4)
        if pend is not None: pend.wait(); pend.stage_ms()
        pend = p
    pend.wait(); ctx.sync()
    print(f"speculative={spec}: {1e3*(time.perf_counter()-t0)/N:.3f} ms per tile")
print("---- with stage events + worker thread (pandas) ----")
from concurrent.futures import ThreadPoolExecutor
pool = ThreadPoolExecutor(max_workers=1)
def host_half(pend):
    raw = pend.wait(); spans = pend.stage_ms(); f = raw.to_frame()
    return pair.score_frame(f, 0.4)
for spec in (1, 0):
    ctx.set_option("speculative", spec)
    N = 30
    t0 = time.perf_counter()
    fut = None
    for i in range(N):
        nxt = pool.submit(host_half, pair.submit_tile(conf, zncc_threshold=0.4))
        if fut is not None: fut.result()
        fut = nxt
    fut.result(); ctx.sync()
    print(f"speculative={spec}: {1e3*(time.perf_counter()-t0)/N:.3f} ms per tile")
