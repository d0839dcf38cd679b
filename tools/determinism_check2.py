#!/usr/bin/env python3
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from karios_amd import synth, ops
from karios_amd.core import KLTConfiguration
S = int(sys.argv[1]) if len(sys.argv) > 1 else 10980
dev = torch.device("cuda", 0)
mon_t, ref_t = synth.make_pair_torch(S, S, 0.5, 0.25, device=dev)
mon = mon_t.cpu().numpy().view(np.uint16); ref = ref_t.cpu().numpy().view(np.uint16)
conf = KLTConfiguration()
def same(rs): return all(np.array_equal(rs[0], r) for r in rs[1:])
A = [ops.klt_tile(ref, mon, conf, mon_ksize=7, ref_ksize=7)[1][0] for _ in range(3)]
print("A klt_tile auto-mask p0 identical:", same(A))
ones = np.ones((S, S), np.uint8)
B = [ops.klt_tile(ref, mon, conf, mask_box=ones, mon_ksize=7, ref_ksize=7)[1][0] for _ in range(3)]
print("B klt_tile user-mask p0 identical:", same(B), "A0==B0", np.array_equal(A[0], B[0]))
lr = ops.laplacian_u8(ops.to_uint8(ref), 7); lm = ops.laplacian_u8(ops.to_uint8(mon), 7)
Cn = [ops.klt_track(lr, lm, None, conf)[0] for _ in range(3)]
print("C klt_track no mask identical:", same(Cn), "C0==A0", np.array_equal(Cn[0], A[0]), "C0==B0", np.array_equal(Cn[0], B[0]))
Cm = [ops.klt_track(lr, lm, ones, conf)[0] for _ in range(3)]
print("D klt_track ones mask identical:", same(Cm), "D0==C0", np.array_equal(Cm[0], Cn[0]))
g = ops.good_features_to_track(lr, 20000, 0.1, 10, blockSize=15)
print("gftt == C0", np.array_equal(g, Cn[0]))
for name, X in (("A", A), ("B", B), ("C", Cn), ("D", Cm)):
    print(name, [int((X[0] != x).any(axis=(1, 2)).sum()) for x in X[1:]], "vs gftt", int((X[0] != g).any(axis=(1,2)).sum()))
