import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from karios_amd import synth
from karios_amd._lib import Context
from karios_amd.core import KLTConfiguration
from karios_amd.resident import ResidentPair
for S in (2000, 5490, 10980):
    dev = torch.device("cuda", 0)
    mon_t, ref_t = synth.make_pair_torch(S, S, 0.5, 0.25, device=dev); torch.cuda.synchronize()
    ctx = Context(0)
    pair = ResidentPair.from_device_pointers(mon_t.data_ptr(), ref_t.data_ptr(), np.uint16, S, S, ctx=ctx)
    for mc in (20000, 2000):
        conf = KLTConfiguration(maxCorners=mc)
        raw = pair.submit_tile(conf, zncc_threshold=0.4).wait()
        print(S, mc, "flags", bin(raw.flags), "rows", raw.n_rows, "cand", raw.n_candidates)
