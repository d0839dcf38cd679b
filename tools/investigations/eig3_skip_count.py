#!/usr/bin/env python3
"""VERDICT r4 item 5, "count before building": in the fused 8-px eigenvalue pass, how many (wavefront, row, pixel slot) triples have
min(S_xx, S_yy) scale^2 <= the running lower bound of the quality threshold in ALL 64 lanes - i.e. could skip the eigenvalue formula
exactly?  Development build (KARIOS_HIP_LIB=karios_amd/libkarios_hip_dev.so), one blocking tile call per workload at 10980^2."""
import ctypes as C
import json
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
assert ctx.lib.km_is_dev_build() == 1, "run with KARIOS_HIP_LIB=karios_amd/libkarios_hip_dev.so (make -C karios_amd/csrc DEV=1)"
ctx.set_option("eig3_count", 1)
out = {}
for name, make, conf in (
        ("config2", lambda: synth.make_pair_torch(S, S, 0.5, 0.25, device=dev), KLTConfiguration()),
        ("config5", lambda: synth.make_cross_sensor_pair_torch(S, S, device=dev)[:2], KLTConfiguration()),
        ("hard_content", lambda: synth.make_hard_pair_torch(S, S, device=dev), KLTConfiguration()),
        ("tie_heavy", lambda: synth.make_tie_heavy_pair_torch(S, S, device=dev), KLTConfiguration()),
        ("e2e_shape_k5", lambda: synth.make_pair_torch(S, S, 0.5, 0.25, device=dev), KLTConfiguration(laplacian_kernel_size=5))):
    mon, ref = make()
    torch.cuda.synchronize()
    pair = ResidentPair.from_device_pointers(mon.data_ptr(), ref.data_ptr(), np.uint16, S, S, ctx=ctx, keepalive=(mon, ref))
    raw = pair.match_tile_raw(conf)
    cnt = (C.c_uint64 * 2)()
    ctx.check(ctx.lib.km_dev_counters(ctx.handle, cnt), "km_dev_counters")
    st = ctx.stats()
    out[name] = {"triples": int(cnt[0]), "skippable": int(cnt[1]), "fraction": round(cnt[1] / max(1, cnt[0]), 4), "corners": int(st.n_init),
                 "candidates": int(st.n_candidates)}
    print(name, out[name], flush=True)
    del pair, mon, ref
    torch.cuda.empty_cache()
print(json.dumps(out))
