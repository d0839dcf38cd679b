#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_forced_paths.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py -q -x -m gpu 2>&1 | tail -3
timeout 300 python tools/fuzz_parity.py --cases 400 --seed 9300 --max-size 900 2>&1 | tail -1
B="--steps 60 --warmup 10 --no-cpu-baseline --no-end-to-end --no-config3 --no-config4 --no-config5 --no-in-flight --no-full-scoring"
for z in 1 0 1 0; do
  if [ $z = 1 ]; then export KARIOS_HIP_LK_STAGE_GENERIC=1; else unset KARIOS_HIP_LK_STAGE_GENERIC; fi
  timeout 300 python bench.py $B 2>/dev/null | tail -1 > gpurun_out/r04_s.json
  python - $z <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r04_s.json").read())
print("generic " if sys.argv[1] == "1" else "buffer  ", "ms_per_step", round(d["ms_per_step"], 4), "median", d["step_spread"]["median_ms"], "lk", d["stage_ms"].get("lk_fwd_bwd"))
PY
done
