#!/usr/bin/env python3
"""GPU probe: forward-backward survival of karios_amd.synth.make_hard_pair_torch at full size for a list of (mix, noise, warp) settings."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from karios_amd import synth
from karios_amd._lib import Context
from karios_amd.core import KLTConfiguration
from karios_amd.resident import ResidentPair

S = int(os.environ.get("SIZE", "10980"))
dev = torch.device("cuda", 0)
ctx = Context(0)
conf = KLTConfiguration()
for spec in sys.argv[1:]:
    mix, noise, warp = (float(v) for v in spec.split(","))
    mon, ref = synth.make_hard_pair_torch(S, S, mix=mix, noise_sigma=noise, warp=warp, device=dev)
    torch.cuda.synchronize()
    pair = ResidentPair.from_device_pointers(mon.data_ptr(), ref.data_ptr(), np.uint16, S, S, ctx=ctx, keepalive=(mon, ref))
    raw = pair.match_tile_raw(conf, zncc_threshold=0.4)
    n_init = int(raw.block[:4].view(np.int32)[1])
    f = raw.to_frame()
    print(f"mix {mix} noise {noise} warp {warp}: {raw.n_rows} of {n_init} = {raw.n_rows / max(1, n_init):.3f}; median dx dy {np.median(f['dx']):.3f} {np.median(f['dy']):.3f}; "
          f"zncc>=0.4 rows {(f['score'] >= 0.4).sum()}", flush=True)
    del pair, mon, ref
    torch.cuda.empty_cache()
