#!/usr/bin/env python3
"""rocprofv3 workload: N blocking tile calls on the headline pair - every kernel of a unit runs ALONE (nothing of the next unit beside
it), so the kernel averages of `rocprofv3 --kernel-trace --stats` are the kernels' own times.  python tools/blocking_workload.py [n] [hard|plain]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from karios_amd import synth
from karios_amd._lib import Context
from karios_amd.core import KLTConfiguration
from karios_amd.resident import ResidentPair

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
kind = sys.argv[2] if len(sys.argv) > 2 else "plain"
S = 10980
dev = torch.device("cuda", 0)
ctx = Context(0)
mon, ref = (synth.make_hard_pair_torch(S, S, device=dev) if kind == "hard" else synth.make_pair_torch(S, S, 0.5, 0.25, device=dev))
torch.cuda.synchronize()
pair = ResidentPair.from_device_pointers(mon.data_ptr(), ref.data_ptr(), np.uint16, S, S, ctx=ctx, keepalive=(mon, ref))
conf = KLTConfiguration()
rows = 0
for _ in range(n):
    rows += pair.match_tile_raw(conf, zncc_threshold=0.4).n_rows
print("ok", rows // n)
