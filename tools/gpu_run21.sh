#!/bin/bash
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_matcher_mirror.py tests/test_gpu_forced_paths.py tests/test_gpu_fuzz.py tests/test_gpu_config4.py -q -x -m gpu > gpurun_out/r04_run21_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r04_run21_tests.log | cut -c1-200
for rep in 1 2; do
for ov in 1 0; do
KARIOS_HIP_TAIL_OVERLAP=$ov python bench.py --no-cpu-baseline --no-end-to-end --no-in-flight --no-config3 --no-config4 --no-config5 --steps 60 --warmup 10 2>/dev/null | tail -1 > gpurun_out/r04_tail_$ov.json
python3 - <<PY
import json
d=json.load(open('gpurun_out/r04_tail_$ov.json'))
print('overlap', $ov, 'ms_per_step', round(d['ms_per_step'],4), 'full', d.get('full_scoring',{}).get('ms_per_pair'), 'gate', d.get('parity_gate',{}).get('passed'), d.get('full_scoring',{}).get('gate',{}).get('passed'))
PY
done
done
