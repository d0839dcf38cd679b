#!/usr/bin/env python3
"""PCIe-inclusive rate of the drop-in path: host numpy images in -> DataFrame out through karios_amd.matcher.KLT
(the 2 x 241 MB upload is inside the timed region).  Never the bench `value`; quoted in DESIGN.md section 7."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from karios_amd import synth
from karios_amd.core import KLTConfiguration, NumpyRasterImage
from karios_amd.matcher import KLT

S = int(sys.argv[1]) if len(sys.argv) > 1 else 10980
mon_t, ref_t = synth.make_pair_torch(S, S, 0.5, 0.25, device=torch.device("cuda", 0))
mon, ref = mon_t.cpu().numpy().view(np.uint16), ref_t.cpu().numpy().view(np.uint16)
del mon_t, ref_t
klt = KLT(KLTConfiguration())
times = []
for it in range(5):
    t0 = time.perf_counter()
    frames = list(klt.match(NumpyRasterImage(mon), NumpyRasterImage(ref), None))
    times.append(time.perf_counter() - t0)
n = sum(len(f) for f in frames)
best = min(times[1:])
print(f"host arrays {S}x{S} uint16 -> {n} key points: {1e3 * best:.1f} ms per pair ({S * S / 1e6 / best:.0f} Mpx/s), all runs ms: {[round(1e3 * t, 1) for t in times]}")
