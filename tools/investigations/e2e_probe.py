#!/usr/bin/env python3
"""The end_to_end object of the bench alone (page-locked host rasters -> KLT.match -> DataFrame + ZNCC), and the same with the
read-only guard of karios_amd.resident.shared_pair switched off: what the guard costs per pair."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from benchkit import legs
from karios_amd import synth, resident
from karios_amd._lib import Context
S = 10980
ctx = Context(0)
mon_t, ref_t = synth.make_pair_torch(S, S, 0.5, 0.25, device=torch.device("cuda", 0)); torch.cuda.synchronize()
mon, ref = mon_t.cpu().numpy().view(np.uint16), ref_t.cpu().numpy().view(np.uint16)
del mon_t, ref_t
for rep in range(2):
    print("guard on ", round(legs.end_to_end(mon, ref, ctx, 10)["ms_per_pair"], 3), flush=True)
    orig, orig_intact = resident._SharedEntry.__init__, resident._SharedEntry.intact
    def no_guard(self, pair, m, r, rasters):
        self.pair, self.mon, self.ref = pair, m, r
        self.raster_ids = tuple(id(x) for x in rasters if x is not None); self.rasters = tuple(x for x in rasters if x is not None); self.guarded = []
    resident._SharedEntry.__init__ = no_guard
    resident._SharedEntry.intact = lambda self, m, r: True
    print("guard off", round(legs.end_to_end(mon, ref, ctx, 10)["ms_per_pair"], 3), flush=True)
    resident._SharedEntry.__init__ = orig
    resident._SharedEntry.intact = orig_intact
    resident.forget_shared_pairs()
