#!/bin/bash
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_forced_paths.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py tests/test_gpu_config4.py -q -x -m gpu 2>&1 | tail -3
for rep in 1 2 3; do
for v in 1 0; do
KARIOS_HIP_DEFER_VALID=$v python bench.py --no-cpu-baseline --no-end-to-end --no-in-flight --no-config3 --no-config4 --no-config5 --no-full-scoring --steps 60 --warmup 10 2>/dev/null | tail -1 > gpurun_out/r04_dv.json
python3 - <<PY
import json
d=json.load(open('gpurun_out/r04_dv.json'))
print('defer_valid', $v, 'ms_per_step', round(d['ms_per_step'],4), d['step_spread']['median_ms'])
PY
done
done
