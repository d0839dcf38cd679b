#!/bin/bash
mkdir -p gpurun_out
B="--steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end --no-config3 --no-config4 --no-config5 --no-in-flight --no-full-scoring"
KARIOS_BENCH_EXCHANGE=1 timeout 300 python bench.py $B > gpurun_out/r04_ex.json 2> gpurun_out/r04_ex.err; echo "rc=$?"; tail -5 gpurun_out/r04_ex.err
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r04_ex.json").read().strip().splitlines()[-1])
    print("bench with exchange", round(d["ms_per_step"], 4), d["step_spread"]["median_ms"], json.dumps(d["exchange"])[:500])
except Exception as e:
    print("no line", e)
PY
timeout 2400 python -m pytest tests -q -x -m gpu > gpurun_out/r04_run9_tests.log 2>&1; echo "tests rc=$?"
tail -5 gpurun_out/r04_run9_tests.log
