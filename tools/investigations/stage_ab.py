#!/usr/bin/env python3
"""Stage spans of BLOCKING tile calls on the headline pair (every kernel alone): median over n calls.  For same-box A/B runs of one
build with different KARIOS_HIP_* development variables (KARIOS_HIP_LIB=karios_amd/libkarios_hip_dev.so).  python tools/investigations/stage_ab.py [n] [hard|plain]"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from karios_amd import synth
from karios_amd._lib import Context
from karios_amd.core import KLTConfiguration
from karios_amd.resident import ResidentPair

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
kind = sys.argv[2] if len(sys.argv) > 2 else "plain"
S = 10980
dev = torch.device("cuda", 0)
ctx = Context(0)
mon, ref = (synth.make_hard_pair_torch(S, S, device=dev) if kind == "hard" else synth.make_pair_torch(S, S, 0.5, 0.25, device=dev))
torch.cuda.synchronize()
pair = ResidentPair.from_device_pointers(mon.data_ptr(), ref.data_ptr(), np.uint16, S, S, ctx=ctx, keepalive=(mon, ref))
conf = KLTConfiguration()
for _ in range(5):
    pair.match_tile_raw(conf, zncc_threshold=0.4)
ctx.set_profiling(True)
ctx.set_option("profile_stage", -1)
acc = {}
for _ in range(n):
    pair.match_tile_raw(conf, zncc_threshold=0.4)
    for k, v in ctx.stage_ms().items():
        acc.setdefault(k, []).append(v)
print({k: round(statistics.median(v), 4) for k, v in acc.items() if max(v) > 0})
