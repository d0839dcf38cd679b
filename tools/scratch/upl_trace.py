import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from karios_amd._lib import Context, pinned_empty
from karios_amd.resident import DeviceBuffer, ResidentPair, shared_pair
ctx = Context(0)
n = 10980 * 10980
a = pinned_empty((10980, 10980), np.uint16, ctx); a[:] = 7
b = pinned_empty((10980, 10980), np.uint16, ctx); b[:] = 9
T = time.perf_counter
bufs = [DeviceBuffer(ctx, a.nbytes) for _ in range(4)]
for i in range(6):
    t0 = T(); bufs[i % 4].upload_image_async(a); t1 = T(); bufs[(i + 1) % 4].upload_image_async(b); t2 = T()
    ctx.lib.km_upload_wait(ctx.handle); t3 = T()
    print(f"raw async: call1 {1e3*(t1-t0):.2f} call2 {1e3*(t2-t1):.2f} wait {1e3*(t3-t2):.2f}")
for i in range(6):
    t0 = T(); p = ResidentPair.upload(a, b, ctx=ctx); t1 = T()
    ctx.lib.km_upload_wait(ctx.handle); t2 = T()
    print(f"ResidentPair.upload {1e3*(t1-t0):.2f} wait {1e3*(t2-t1):.2f}")
    del p
for i in range(6):
    t0 = T(); p = ResidentPair.upload(a, b, ctx=ctx); t1 = T(); shared_pair(a, b, ctx, publish=p); t2 = T()
    ctx.lib.km_upload_wait(ctx.handle); t3 = T()
    print(f"upload {1e3*(t1-t0):.2f} publish {1e3*(t2-t1):.2f} wait {1e3*(t3-t2):.2f}")
