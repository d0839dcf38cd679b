#!/usr/bin/env python3
"""Double-precision phase correlation (k_fft64.hip) on the GPU box: answers against the oracle on a list of shapes (smooth sides, the
61 of Sentinel-2, other primes with a level kernel, primes that need Bluestein, 1-D cases) and its time at 10980 x 10980."""
import json
import sys
import time

import numpy as np

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from karios_amd import synth                                             # noqa: E402
import karios_amd.ops as ops                                             # noqa: E402
from karios_amd._lib import default_context                             # noqa: E402


def main():
    from oracle import oracle as O
    ctx = default_context()
    ctx.set_option("phase_fp64", 1)
    shapes = [] if len(sys.argv) > 1 and sys.argv[1] == 'time' else [(64, 64), (96, 130), (61, 45), (128, 128), (244, 183), (366, 366), (122, 3721), (2135, 128), (1, 300), (300, 1), (2, 2), (3, 5),
              (11, 13), (127, 254), (131, 200), (200, 131), (257, 263), (1000, 1009), (1098, 1220), (4096, 64), (64, 4096), (2048, 2048),
              (4099, 37), (37, 4099), (512, 6000), (6000, 512), (3001, 3001)]
    bad = 0
    for (H, W) in shapes:
        base, _ = synth.make_pair(H + 80, W + 80, 0.0, 0.0, seed=H * 3 + W, noise_sigma=0.0)
        rng = np.random.default_rng(H * 7 + W)
        sy = int(rng.integers(-min(30, H // 8), min(30, H // 8) + 1))
        sx = int(rng.integers(-min(30, W // 8), min(30, W // 8) + 1))
        a = base[40:40 + H, 40:40 + W]
        b = base[40 - sy:40 - sy + H, 40 - sx:40 - sx + W]
        t0 = time.perf_counter()
        got = ops.phase_cross_correlation(b, a)
        t1 = time.perf_counter()
        want = O.phase_cross_correlation(b, a)
        ok = np.array_equal(got, want)
        bad += not ok
        print(f"{H:6d} x {W:6d}  shift ({sy:4d},{sx:4d})  gpu {got}  oracle {want}  path {ctx.phase_info()[0]}  {'ok' if ok else 'MISMATCH'}  {1e3 * (t1 - t0):8.1f} ms", flush=True)
    # full size
    from karios_amd.resident import ResidentPair
    H = W = 10980
    mon, ref = synth.make_pair(H, W, 0.0, 0.0, seed=5, noise_sigma=2.0)
    mon = np.roll(ref, (-21, 37), (0, 1))
    pair = ResidentPair.upload(mon, ref)
    res = {}
    for mode, opts in ((1, {}), (1, {"f64_half": 0}), (1, {}), (1, {"f64_half": 0}), (1, {"f64_pair": 0}), (0, {})):
        ctx.set_option("phase_fp64", mode)
        ctx.set_option("f64_plain", 0)
        ctx.set_option("f64_pair", 1)
        ctx.set_option("f64_half", 1)
        for k, v in opts.items():
            ctx.set_option(k, v)
        times = []
        for i in range(4):
            ctx.sync()
            t0 = time.perf_counter()
            got = pair.phase_offset()
            ctx.sync()
            times.append(1e3 * (time.perf_counter() - t0))
        res[("fp64" if mode else "f32") + "".join(f" {k}={v}" for k, v in opts.items())] = {"ms": times, "shift": None if got is None else [float(v) for v in got], "path": ctx.phase_info()[0]}
        print(mode, opts, [round(t, 2) for t in times], got, ctx.phase_info(), flush=True)
    print(json.dumps({"mismatches": bad, "full_size": res}))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
