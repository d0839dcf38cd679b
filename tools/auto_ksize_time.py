#!/usr/bin/env python3
"""Dev tool: wall time of the batched auto-ksize search (km_klt_auto_ksize_frame_dev) on a resident S2-sized pair."""
import os, sys, time
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
conf = KLTConfiguration(laplacian_kernel_size="auto")
for it in range(3):
    t0 = time.perf_counter()
    frame, scores, best, ninit = pair.match_tile_auto_ksize(conf)
    dt = time.perf_counter() - t0
    print(f"auto-ksize search {S}x{S}: {1e3 * dt:.1f} ms, best {best}, {len(frame)} key points of {ninit}, "
          f"ratios {min(scores.values()):.3f}..{max(scores.values()):.3f}")
