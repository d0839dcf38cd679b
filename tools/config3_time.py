#!/usr/bin/env python3
"""Dev tool: timing of the large-shift path (BASELINE config 3) on a resident 10980^2 pair."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from karios_amd import synth
from karios_amd._lib import Context
from karios_amd.core import KLTConfiguration
from karios_amd.resident import ResidentPair
S = int(sys.argv[1]) if len(sys.argv) > 1 else 10980
dev = torch.device("cuda", 0)
mon_t, ref_t = synth.make_pair_torch(S, S, 37.25, -20.75, device=dev)
torch.cuda.synchronize()
ctx = Context(0)
pair = ResidentPair.from_device_pointers(mon_t.data_ptr(), ref_t.data_ptr(), np.uint16, S, S, ctx=ctx)
conf = KLTConfiguration()
for it in range(4):
    t0 = time.perf_counter(); off = pair.phase_offset(); t1 = time.perf_counter()
    sh = pair.shifted_monitored(int(off[0]), int(off[1])); ctx.sync(); t2 = time.perf_counter()
    fr = sh.match_tile(conf); t3 = time.perf_counter()
    print(f"phase_offset={1e3*(t1-t0):.2f}ms shift={1e3*(t2-t1):.2f}ms klt={1e3*(t3-t2):.2f}ms offsets={off} n={len(fr)} med=({np.median(fr['dx'])+off[1]:.3f},{np.median(fr['dy'])+off[0]:.3f})")
    del sh
