#!/bin/bash
for rep in 1 2; do
for at in 0 1 2; do
KARIOS_HIP_MM_EARLY_AT=$at python bench.py --no-cpu-baseline --no-end-to-end --no-in-flight --no-config3 --no-config4 --no-config5 --no-full-scoring --steps 60 --warmup 10 2>/dev/null | tail -1 > gpurun_out/r04_mm_$at.json
python3 - <<PY
import json
d=json.load(open('gpurun_out/r04_mm_$at.json'))
print('mm_early_at', $at, 'ms_per_step', round(d['ms_per_step'],4), 'gate', d.get('parity_gate',{}).get('passed'), {k: round(v,3) for k,v in d.get('stage_ms',{}).items()})
PY
done
done
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "scoring_tail or submit_wait" 2>&1 | tail -3
