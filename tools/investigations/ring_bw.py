#!/usr/bin/env python3
"""Upload rate of a PAGEABLE 10980^2 uint16 raster through the library's staging ring (csrc/staging.hip) against the page-locked
rate: `KARIOS_HIP_COPY_THREADS=N python tools/investigations/ring_bw.py` (the pool's size is read once per process)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from karios_amd import pinned_empty
from karios_amd._lib import Context
from karios_amd.resident import DeviceBuffer

S = 10980
ctx = Context(0)
a = (np.arange(S * S, dtype=np.uint32) % 65521).astype(np.uint16).reshape(S, S)
p = pinned_empty((S, S), np.uint16, ctx)
p[...] = a
buf = DeviceBuffer(ctx, a.nbytes)
for name, src in (("pageable", a), ("page-locked", p)):
    best = 1e9
    for _ in range(4):
        t0 = time.perf_counter()
        buf.upload_image_async(src)
        ctx.lib.km_upload_wait(ctx.handle)
        best = min(best, time.perf_counter() - t0)
    back = buf.download((S, S), np.uint16)
    assert np.array_equal(back, a)
    print(f"copy threads {os.environ.get('KARIOS_HIP_COPY_THREADS', 'default')}: {name} {a.nbytes / best / 1e9:.1f} GB/s ({best * 1e3:.2f} ms for {a.nbytes >> 20} MiB)")
