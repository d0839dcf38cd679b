#!/usr/bin/env python3
"""Dev tool: run the full-size pipeline stages twice and report which outputs differ between runs."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from karios_amd import synth, ops
from karios_amd.core import KLTConfiguration
from karios_amd.resident import ResidentPair

S = int(sys.argv[1]) if len(sys.argv) > 1 else 10980
dev = torch.device("cuda", 0)
mon_t, ref_t = synth.make_pair_torch(S, S, 0.5, 0.25, device=dev)
torch.cuda.synchronize()
pair = ResidentPair.from_device_pointers(mon_t.data_ptr(), ref_t.data_ptr(), np.uint16, S, S)
conf = KLTConfiguration()
runs = [pair.track_tile(conf)[1] for _ in range(3)]
for name, i in (("p0", 0), ("p1", 1), ("p0r", 2)):
    print(name, "identical:", all(np.array_equal(runs[0][i], r[i]) for r in runs[1:]))
ref = ref_t.cpu().numpy().view(np.uint16)
u8 = ops.to_uint8(ref)
lap = [ops.laplacian_u8(u8, 7) for _ in range(2)]
print("lap identical:", np.array_equal(lap[0], lap[1]))
eig = [ops.min_eigen(lap[0], 15) for _ in range(2)]
print("eig identical:", np.array_equal(eig[0].view(np.uint32), eig[1].view(np.uint32)), "n diff", int((eig[0].view(np.uint32) != eig[1].view(np.uint32)).sum()))
g = [ops.good_features_to_track(lap[0], 20000, 0.1, 10, blockSize=15) for _ in range(3)]
print("gftt identical:", all(np.array_equal(g[0], x) for x in g[1:]), [len(x) for x in g])
if not np.array_equal(g[0], g[1]):
    a = {tuple(p) for p in g[0].reshape(-1, 2)}; b = {tuple(p) for p in g[1].reshape(-1, 2)}
    print("common corners", len(a & b), "first mismatch index", int(np.argmax((g[0] != g[1]).any(axis=(1, 2)))))
