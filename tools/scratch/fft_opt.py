#!/usr/bin/env python3
"""Dev tool: phase-correlation time with a km_set_option knob toggled (same process, same box)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from karios_amd import synth
from karios_amd._lib import Context
from karios_amd.resident import ResidentPair
S = 10980
knob = sys.argv[1] if len(sys.argv) > 1 else "fft_cross"
dev = torch.device("cuda", 0)
mon_t, ref_t = synth.make_pair_torch(S, S, 37.25, -20.75, device=dev)
torch.cuda.synchronize()
ctx = Context(0)
pair = ResidentPair.from_device_pointers(mon_t.data_ptr(), ref_t.data_ptr(), np.uint16, S, S, ctx=ctx)
ctx.set_profiling(True)
for rep in range(2):
    for v in (1, 0):
        ctx.set_option(knob, v)
        ts = []
        for _ in range(8):
            off = pair.phase_offset()
            ts.append(ctx.stage_ms().get("phase_correlation", 0.0))
        print(f"{knob}={v} phase ms min {min(ts):.3f} median {sorted(ts)[4]:.3f}", off, ctx.phase_info())
