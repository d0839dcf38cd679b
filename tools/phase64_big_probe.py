import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from karios_amd._lib import default_context
from karios_amd.resident import ResidentPair
ctx = default_context()
rng = np.random.default_rng(3)
for (H, W) in ((20000, 20000), (15000, 12289), (9973, 10007)):
    ref = rng.integers(0, 4000, (H, W), dtype=np.uint16)
    # smooth a little so that the correlation peak is not a delta of noise only
    mon = np.roll(ref, (-33, 58), (0, 1))
    pair = ResidentPair.upload(mon, ref)
    for i in range(2):
        ctx.sync(); t0 = time.perf_counter(); got = pair.phase_offset(); ctx.sync(); dt = 1e3 * (time.perf_counter() - t0)
    print(H, W, got, ctx.phase_info(), round(dt, 2), "ms", flush=True)
    assert tuple(got) == (-33.0, 58.0), got
    del pair
print("ok")
