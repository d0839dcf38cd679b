#!/bin/bash
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -x -m gpu > gpurun_out/r04_run16_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r04_run16_tests.log
timeout 300 python tools/fuzz_parity.py --cases 300 --seed 9900 --max-size 900 --force-paths 2>&1 | tail -1
B="--steps 60 --warmup 10 --no-cpu-baseline --no-end-to-end --no-config3 --no-config4 --no-config5 --no-in-flight --no-full-scoring"
for z in 1 0 1 0; do
  if [ $z = 1 ]; then export KARIOS_HIP_FRAME_TWO_PASS=1; else unset KARIOS_HIP_FRAME_TWO_PASS; fi
  timeout 300 python bench.py $B 2>/dev/null | tail -1 > gpurun_out/r04_f.json
  python - $z <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r04_f.json").read())
print("two launches" if sys.argv[1] == "1" else "one launch  ", "ms_per_step", round(d["ms_per_step"], 4), "median", d["step_spread"]["median_ms"], "frame", d["stage_ms"].get("fb_frame"))
PY
done
