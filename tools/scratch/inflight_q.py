#!/usr/bin/env python3
"""Dev tool: three pairs in flight vs hardware queues / aux stream (run with GPU_MAX_HW_QUEUES set by the caller)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from karios_amd import synth
from karios_amd.core import KLTConfiguration
dev = torch.device("cuda", 0)
S = 10980
first = synth.make_pair_torch(S, S, 0.5, 0.25, device=dev)
torch.cuda.synchronize()
for n_ctx in (1, 2, 3):
    r = bench.in_flight(dev, KLTConfiguration(), S, first, n_ctx=n_ctx, pairs=60)
    print(os.environ.get("GPU_MAX_HW_QUEUES"), os.environ.get("KARIOS_HIP_OPTIONS"), "n_ctx", n_ctx, round(r["ms_per_pair"], 4), r["tiles_redone"])
