#!/usr/bin/env python3
"""Dev tool: stage times of one resident tile match, whatever the result (used with experimental library builds)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from karios_amd import synth
from karios_amd._lib import Context
from karios_amd.core import KLTConfiguration
from karios_amd.resident import ResidentPair

S = int(sys.argv[1]) if len(sys.argv) > 1 else 10980
mon_t, ref_t = synth.make_pair_torch(S, S, 0.5, 0.25, device=torch.device("cuda", 0))
torch.cuda.synchronize()
ctx = Context(0)
pair = ResidentPair.from_device_pointers(mon_t.data_ptr(), ref_t.data_ptr(), np.uint16, S, S, ctx=ctx)
conf = KLTConfiguration()
ctx.set_profiling(True)
for _ in range(3):
    try:
        pair.match_tile(conf)
    except Exception as e:  # noqa: BLE001 - experimental builds may produce nonsense
        print("match_tile:", type(e).__name__, e)
    print({k: round(v, 3) for k, v in ctx.stage_ms().items()})
