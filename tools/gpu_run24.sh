#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_matcher_mirror.py tests/test_gpu_golden.py tests/test_gpu_parity.py -q -x -m gpu -k "phase or large_offset or config3" 2>&1 | tail -4
python - <<'PY'
import sys, time, numpy as np
sys.path.insert(0, '.')
from karios_amd import synth
from karios_amd._lib import default_context
from karios_amd.resident import ResidentPair
ctx = default_context()
_, ref = synth.make_pair(10980, 10980, 0.0, 0.0, seed=5, noise_sigma=2.0)
mon = np.roll(ref, (-21, 37), (0, 1))
pair = ResidentPair.upload(mon, ref)
for herm in (1, 0, 1, 0):
    ctx.set_option("fft_herm", herm)
    ts = []
    for i in range(5):
        ctx.sync(); t0 = time.perf_counter(); got = pair.phase_offset(); ctx.sync(); ts.append(1e3 * (time.perf_counter() - t0))
    print("fft_herm", herm, [round(t, 3) for t in ts], got, ctx.phase_info())
PY
python bench.py --config 3 --steps 10 --warmup 2 2> gpurun_out/bench_r04_config3.err | tail -1 > gpurun_out/bench_r04_config3.json; python3 - <<'PY'
import json
d=json.load(open('gpurun_out/bench_r04_config3.json'))
print(d['ms_per_step'], d['stage_ms'], d.get('phase_fp64',{}).get('ms'), d.get('gate',{}).get('passed'), d['roofline']['frac'], d['phase_path'])
PY
