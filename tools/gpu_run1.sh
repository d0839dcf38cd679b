#!/bin/bash
# round 4, GPU call 1: new parity tests (staging ring, oscillation literal), the upload control experiment, a first bench line
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu > gpurun_out/r04_run1_parity.log 2>&1; echo "parity rc=$?" 
tail -3 gpurun_out/r04_run1_parity.log
timeout 700 python tools/upload_stress.py --mode old --seconds 420 > gpurun_out/r04_stress_old.log 2>&1; echo "stress old rc=$?"
tail -5 gpurun_out/r04_stress_old.log
timeout 400 python tools/upload_stress.py --mode ring --seconds 150 > gpurun_out/r04_stress_ring.log 2>&1; echo "stress ring rc=$?"
tail -3 gpurun_out/r04_stress_ring.log
timeout 900 python bench.py --steps 20 --warmup 5 --no-config3 --no-config4 --no-config5 --no-in-flight --no-cpu-baseline > gpurun_out/r04_bench_a.json 2> gpurun_out/r04_bench_a.err; echo "bench rc=$?"
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r04_bench_a.json").read().strip().splitlines()[-1])
    print("ms_per_step", d["ms_per_step"], "value", d["value"])
    print("end_to_end", json.dumps(d.get("end_to_end"))[:600])
except Exception as e:
    print("bench parse failed", e)
PY
