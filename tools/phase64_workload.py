#!/usr/bin/env python3
"""rocprofv3 workload: the double-precision phase correlation (k_fft64.hip) of a resident 10980 x 10980 uint16 pair, `reps` times."""
import sys

import numpy as np

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from karios_amd import synth                                             # noqa: E402
from karios_amd._lib import default_context                             # noqa: E402
from karios_amd.resident import ResidentPair                            # noqa: E402

side = int(sys.argv[1]) if len(sys.argv) > 1 else 10980
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ctx = default_context()
ctx.set_option("phase_fp64", 1)
for kv in os.environ.get("KARIOS_OPTS", "").split(","):                 # development library: name=value,...
    if kv:
        ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
_, ref = synth.make_pair(side, side, 0.0, 0.0, seed=5, noise_sigma=2.0)
mon = np.roll(ref, (-21, 37), (0, 1))
pair = ResidentPair.upload(mon, ref)
for _ in range(reps):
    got = pair.phase_offset()
assert os.environ.get("KARIOS_TIMING_ONLY") or tuple(got) == (-21.0, 37.0), got   # (KARIOS_TIMING_ONLY: experimental libraries with wrong arithmetic)
print("ok", got)
