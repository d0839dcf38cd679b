import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from karios_amd import synth, pinned_empty
from karios_amd._lib import Context
from karios_amd.core import KLTConfiguration, NumpyRasterImage
from karios_amd.matcher import KLT, ZNCCService
S = 10980
dev = torch.device("cuda", 0)
mon_t, ref_t = synth.make_pair_torch(S, S, 0.5, 0.25, device=dev); torch.cuda.synchronize()
mon, ref = mon_t.cpu().numpy().view(np.uint16), ref_t.cpu().numpy().view(np.uint16)
del mon_t, ref_t
ctx = Context(0)
conf = KLTConfiguration()
pairs = []
for k in range(2):
    pm, pr = pinned_empty(mon.shape, mon.dtype, ctx), pinned_empty(ref.shape, ref.dtype, ctx)
    np.copyto(pm, mon); np.copyto(pr, ref)
    pairs.append((NumpyRasterImage(pm), NumpyRasterImage(pr)))
klt, zncc = KLT(conf, ctx=ctx), ZNCCService(ctx=ctx)
klt.prefetch(*pairs[0], None)
T = time.perf_counter
for i in range(8):
    cur, nxt = pairs[i % 2], pairs[(i + 1) % 2]
    t0 = T(); gen = klt.match(cur[0], cur[1], None)
    klt.prefetch(nxt[0], nxt[1], None); t1 = T()
    f = next(gen); t2 = T()
    rest = list(gen); t3 = T()
    cand = f[f["score"] >= 0.4]; z = zncc.compute_zncc(cand, cur[0], cur[1]); t4 = T()
    ctx.lib.km_upload_wait(ctx.handle); t5 = T()
    print(f"iter {i}: prefetch {1e3*(t1-t0):.2f}  first frame {1e3*(t2-t1):.2f}  gen end {1e3*(t3-t2):.2f}  zncc {1e3*(t4-t3):.2f}  upload_wait {1e3*(t5-t4):.2f}  total {1e3*(t5-t0):.2f}")
print("---- instrumented")
import karios_amd.resident as R, karios_amd._lib as L, karios_amd.matcher.klt as K
def wrap(obj, name, label):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = T(); r = f(*a, **k); dt = 1e3 * (T() - t0)
        if dt > 0.5: print(f"      {label} {dt:.2f} ms")
        return r
    setattr(obj, name, g)
wrap(L.Context, "dev_release", "dev_release"); wrap(L.Context, "dev_alloc", "dev_alloc")
wrap(R.DeviceBuffer, "upload_image_async", "upload_image_async")
wrap(R, "_identity", "_identity"); wrap(K, "shared_pair", "shared_pair(publish)")
for i in range(3):
    cur, nxt = pairs[i % 2], pairs[(i + 1) % 2]
    t0 = T(); gen = klt.match(cur[0], cur[1], None)
    klt.prefetch(nxt[0], nxt[1], None); t1 = T()
    f = next(gen); list(gen); t2 = T()
    cand = f[f["score"] >= 0.4]; z = zncc.compute_zncc(cand, cur[0], cur[1]); t4 = T()
    print(f"iter {i}: prefetch {1e3*(t1-t0):.2f}  match {1e3*(t2-t1):.2f}  zncc {1e3*(t4-t2):.2f}")
