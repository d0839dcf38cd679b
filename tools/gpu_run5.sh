#!/bin/bash
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -x -m gpu > gpurun_out/r04_run5_tests.log 2>&1; echo "tests rc=$?"
tail -4 gpurun_out/r04_run5_tests.log
timeout 900 python tools/exchange_probe.py 150 --json 2>/dev/null | tail -1 | cut -c1-700
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_d.json 2> gpurun_out/r04_bench_d.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04_bench_d.json").read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], "median", d["step_spread"]["median_ms"], "stage", d["stage_ms"])
print("full_scoring", d["full_scoring"]["ms_per_pair"], d["full_scoring"]["parity"]["passed"], "e2e", d["end_to_end"]["ms_per_pair"], d["end_to_end"]["pageable_numpy_ms_per_pair"])
print("config3", d["config3"]["ms_per_step"], "config4", d["config4"]["ms_per_step"], "config5", d["config5"]["ms_per_pair"] if "ms_per_pair" in d["config5"] else d["config5"].get("ms_per_step"), "in_flight", d["in_flight"]["ms_per_pair"])
print("parity", d["cpu_baseline"]["parity"]["passed"], "cpu", d["cpu_baseline"]["value"])
PY
