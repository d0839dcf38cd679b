#!/usr/bin/env python3
"""The config-4 object of the bench alone (16 units of 5490^2 per step, batched, steps pipelined): ms per step.
python tools/config4_probe.py [steps, default 8]      (KARIOS_C4_TRACE=1: host-side times of every submit / collect)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from benchkit.config4 import config4

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
for rep in range(2):
    r = config4(dev, 0, 1, dev, steps, batched=True)
    print("batched", round(r["ms_per_step"], 3), r["units_gathered"], r["matched_keypoints_per_step"], flush=True)
