#!/bin/bash
timeout 2400 python -m pytest tests -q -x -m gpu --durations=8 > gpurun_out/r04_run20_tests.log 2>&1; echo "tests rc=$?"; tail -14 gpurun_out/r04_run20_tests.log | cut -c1-200
python bench.py --config 3 --steps 10 --warmup 2 2> gpurun_out/bench_r04_config3.err | tail -1 > gpurun_out/bench_r04_config3.json; python3 - <<'PY'
import json
d=json.load(open('gpurun_out/bench_r04_config3.json'))
print(d['ms_per_step'], d['stage_ms'], d.get('phase_fp64'), d.get('gate',{}).get('passed'))
PY
