#!/bin/bash
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -x -m gpu > gpurun_out/r04_run13_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r04_run13_tests.log
timeout 300 python tools/fuzz_parity.py --what aux --cases 200 --seed 8100 2>&1 | tail -1
timeout 300 python tools/fuzz_parity.py --cases 150 --seed 8300 --max-size 900 2>&1 | tail -1
for z in 1 0 1 0; do
  if [ $z = 1 ]; then export KARIOS_HIP_ZNCC_TWO_PASS=1; else unset KARIOS_HIP_ZNCC_TWO_PASS; fi
  timeout 600 python bench.py --steps 20 --warmup 5 --no-end-to-end --no-config3 --no-config4 --no-config5 --no-in-flight --cpu-runs 1 2>/dev/null | tail -1 > gpurun_out/r04_z.json
  python - $z <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r04_z.json").read())
fs = d["full_scoring"]
print("two-pass" if sys.argv[1] == "1" else "integer ", "headline", round(d["ms_per_step"], 4), "median", d["step_spread"]["median_ms"], "zncc", d["stage_ms"].get("zncc"), "| full_scoring", round(fs["ms_per_pair"], 4), "mi", fs["stage_ms"].get("mutual_info"), "gate", d["cpu_baseline"]["parity"]["passed"], d["cpu_baseline"]["parity"]["max_abs_dzncc"], fs["parity"]["passed"])
PY
done
