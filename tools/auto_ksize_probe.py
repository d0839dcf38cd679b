#!/usr/bin/env python3
"""The auto_ksize object of the bench line alone (benchkit.legs.auto_ksize_object): timing + the in-run oracle gate."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from benchkit import legs
from karios_amd import synth
from karios_amd._lib import Context
from karios_amd.resident import ResidentPair
S = int(sys.argv[1]) if len(sys.argv) > 1 else 10980
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
ctx = Context(0)
data = synth.make_pair_torch(S, S, 0.5, 0.25, device=dev); torch.cuda.synchronize()
pair = ResidentPair.from_device_pointers(data[0].data_ptr(), data[1].data_ptr(), np.uint16, S, S, ctx=ctx, keepalive=data)
print(json.dumps(legs.auto_ksize_object(ctx, pair, data, S)))
