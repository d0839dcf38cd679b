#!/usr/bin/env python3
"""`KLT.match` on rasters resident in HBM in the reference's end-to-end shape (tile_size 6000, k = 5: four unequal tiles): the grid as ONE
batched submission against two pipelined halves (the first half's DataFrames built while the second half is on the device)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from karios_amd import synth
from karios_amd._lib import Context
from karios_amd.core import KLTConfiguration
from karios_amd.core.image import DeviceRasterImage
from karios_amd.matcher import KLT
from karios_amd.resident import ResidentPair

S = 10980
dev = torch.device("cuda", 0)
ctx = Context(0)
mon_t, ref_t = synth.make_pair_torch(S, S, 0.5, 0.25, device=dev)
torch.cuda.synchronize()
mon_img, ref_img = DeviceRasterImage(mon_t, np.uint16), DeviceRasterImage(ref_t, np.uint16)
orig = ResidentPair.match_pipelined
for tile_size in (6000, 5490, 3000):
    klt = KLT(KLTConfiguration(tile_size=tile_size, laplacian_kernel_size=5), ctx=ctx)
    ref_frames = None
    for rep in range(2):
        for split in (False, True):
            ResidentPair.match_pipelined = lambda self, *a, _s=split, **k: orig(self, *a, split_small_grids=_s, **k)
            frames = list(klt.match(mon_img, ref_img, None))
            ctx.sync()
            if ref_frames is None:
                ref_frames = frames
            assert len(frames) == len(ref_frames) and all(a.equals(b) for a, b in zip(frames, ref_frames))
            w = []
            for _ in range(3):
                t0 = time.perf_counter()
                for _ in range(6):
                    frames = list(klt.match(mon_img, ref_img, None))
                ctx.sync()
                w.append((time.perf_counter() - t0) / 6 * 1e3)
            print(f"tile_size {tile_size} ({len(frames)} tiles) {'two halves' if split else 'one submission'}: {sorted(w)[1]:.3f} ms per pair {[round(v, 3) for v in w]}", flush=True)
ResidentPair.match_pipelined = orig
