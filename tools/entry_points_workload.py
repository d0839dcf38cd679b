#!/usr/bin/env python3
"""Workload for rocprofv3 (kernel trace / --pmc passes): every device-side entry point of the library that the headline loop
does not reach, at Sentinel-2 size, a few calls each - so that every exported entry point has one measured line
(profiles/r04_kernel_stats_entry_points.md, profiles/pmc_traffic.json).

    full scoring   FrameStream(mutual_info=True): KLT + ZNCC + mutual_info_score + mi_score   (core.py:894-907)   -> mi_kernel
    dn filter      results.filter_by_dn_values                                                 (core.py:650-737)   -> dn_keep_kernel
    auto ksize     ResidentPair.match_tile_auto_ksize                                          (klt.py:465-545)
    banded tile    parallel.match_tile_banded on one rank                                      
    phase + shift  ResidentPair.phase_offset / shifted_monitored                               (large_offset.py:39, image.py:70-101)

    python tools/entry_points_workload.py [size] [what,what,...]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from karios_amd import results, synth  # noqa: E402
from karios_amd._lib import Context  # noqa: E402
from karios_amd.core import KLTConfiguration, NumpyRasterImage  # noqa: E402
from karios_amd.resident import ResidentPair  # noqa: E402
from karios_amd.stream import FrameStream  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 10980
what = set((sys.argv[2] if len(sys.argv) > 2 else "scoring,dn,auto,banded,phase").split(","))
dev = torch.device("cuda", 0)
mon_t, ref_t = synth.make_pair_torch(S, S, 0.5, 0.25, device=dev)
torch.cuda.synchronize()
ctx = Context(0)
pair = ResidentPair.from_device_pointers(mon_t.data_ptr(), ref_t.data_ptr(), np.uint16, S, S, ctx=ctx, keepalive=(mon_t, ref_t))
conf = KLTConfiguration()
frame = None
if "scoring" in what:
    with FrameStream(0.4, depth=1, mutual_info=True) as s:
        t0 = time.perf_counter()
        done = []
        for _ in range(6):
            done += s.submit(pair, conf)
        done += s.drain()
        ctx.sync()
        frame = done[-1].frame
        print(f"full scoring: {(time.perf_counter() - t0) / 6 * 1e3:.3f} ms per pair, {len(frame)} rows, columns {list(frame.columns)}")
if frame is None:
    frame = pair.match_tile(conf, zncc_threshold=0.4)
if "dn" in what:
    for _ in range(3):
        t0 = time.perf_counter()
        kept = results.filter_by_dn_values(frame, pair, no_values=[0, 65535])
        print(f"dn filter: {(time.perf_counter() - t0) * 1e3:.3f} ms, {len(kept)} of {len(frame)} rows kept")
if "auto" in what:
    c2 = KLTConfiguration(laplacian_kernel_size="auto")
    for _ in range(2):
        t0 = time.perf_counter()
        f, scores, best, ninit = pair.match_tile_auto_ksize(c2)
        print(f"auto-ksize search: {(time.perf_counter() - t0) * 1e3:.1f} ms, best {best}, {len(f)} rows")
if "banded" in what:
    from karios_amd.parallel import match_tile_banded
    mon, ref = mon_t.cpu().numpy().view(np.uint16), ref_t.cpu().numpy().view(np.uint16)
    for _ in range(2):
        t0 = time.perf_counter()
        got = match_tile_banded(NumpyRasterImage(mon), NumpyRasterImage(ref), None, conf, zncc_threshold=0.4, device="cpu")
        print(f"banded tile (one rank, upload included): {(time.perf_counter() - t0) * 1e3:.1f} ms, {len(got)} rows")
if "phase" in what:
    for _ in range(3):
        t0 = time.perf_counter()
        off = pair.phase_offset()
        sh = pair.shifted_monitored(int(off[0]), int(off[1]))
        ctx.sync()
        print(f"phase correlation + shift: {(time.perf_counter() - t0) * 1e3:.2f} ms, offset {off}, path {ctx.phase_info()}")
        del sh
ctx.sync()
