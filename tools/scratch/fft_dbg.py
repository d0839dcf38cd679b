#!/usr/bin/env python3
"""Dev tool: phase-correlation stage time with phases of the 61 M row kernel skipped (km_set_option fft_dbg)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from karios_amd import synth
from karios_amd._lib import Context
from karios_amd.resident import ResidentPair
S = 10980
dev = torch.device("cuda", 0)
mon_t, ref_t = synth.make_pair_torch(S, S, 37.25, -20.75, device=dev)
torch.cuda.synchronize()
ctx = Context(0)
pair = ResidentPair.from_device_pointers(mon_t.data_ptr(), ref_t.data_ptr(), np.uint16, S, S, ctx=ctx)
ctx.set_profiling(True)
for dbg in (16, 17, 18, 19, 20, 24, 31):
    ctx.set_option("fft_dbg", dbg)
    ts = []
    for _ in range(3):
        try:
            pair.phase_offset()
        except Exception as e:
            pass
        ts.append(ctx.stage_ms().get("phase_correlation", 0.0))
    print(f"dbg={dbg:2d} (skip: {'A ' if dbg&1 else ''}{'B ' if dbg&2 else ''}{'prefetch ' if dbg&4 else ''}{'store' if dbg&8 else ''}) phase={min(ts):.3f} ms")
