#!/usr/bin/env python3
"""Dev tool: wall-clock breakdown of one bench step (host glue vs device pipeline)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from karios_amd import synth
from karios_amd._lib import Context
from karios_amd.core import KLTConfiguration
from karios_amd.resident import ResidentPair

S = int(sys.argv[1]) if len(sys.argv) > 1 else 10980
dev = torch.device("cuda", 0)
mon_t, ref_t = synth.make_pair_torch(S, S, 0.5, 0.25, device=dev)
torch.cuda.synchronize()
ctx = Context(0)
pair = ResidentPair.from_device_pointers(mon_t.data_ptr(), ref_t.data_ptr(), np.uint16, S, S, ctx=ctx)
conf = KLTConfiguration()
for it in range(6):
    t = [time.perf_counter()]
    frame = pair.match_tile(conf, zncc_threshold=0.4); t.append(time.perf_counter())
    frame = pair.score_frame(frame, 0.4); t.append(time.perf_counter())
    if it >= 2:
        names = ["match_tile(device frame)", "score_frame(zncc)"]
        print("  ".join(f"{n}={1e3*(b-a):.3f}ms" for n, a, b in zip(names, t, t[1:])), f"total={1e3*(t[-1]-t[0]):.3f}ms", len(frame))
ctx.set_profiling(True)
pair.match_tile(conf)
print({k: round(v, 3) for k, v in ctx.stage_ms().items()})

st = ctx.stats(); print("candidates", st.n_candidates, "emitted ratio", st.emitted_ratio, "select rounds", st.n_select_batches)
