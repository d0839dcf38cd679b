#!/usr/bin/env python3
"""Dev tool: stage times of one resident full-size tile call (works even when no corner is produced)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from karios_amd import synth
from karios_amd._lib import Context
from karios_amd.core import KLTConfiguration
from karios_amd.resident import ResidentPair
S = int(sys.argv[1]) if len(sys.argv) > 1 else 10980
dev = torch.device("cuda", 0)
mon_t, ref_t = synth.make_pair_torch(S, S, 0.5, 0.25, device=dev)
torch.cuda.synchronize()
ctx = Context(0)
pair = ResidentPair.from_device_pointers(mon_t.data_ptr(), ref_t.data_ptr(), np.uint16, S, S, ctx=ctx)
conf = KLTConfiguration()
for _ in range(3):
    pair.track_tile(conf)
ctx.set_profiling(True)
pair.track_tile(conf)
print({k: round(v, 3) for k, v in ctx.stage_ms().items() if v})
st = ctx.stats(); print("candidates", st.n_candidates, "emitted ratio", st.emitted_ratio)
