#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_golden.py tests/test_gpu_matcher_mirror.py tests/test_gpu_fuzz.py -q -x -m gpu 2>&1 | tail -4
timeout 300 python tools/fuzz_parity.py --what aux --cases 300 --seed 7100 2>&1 | tail -2
for rep in 1 2; do
timeout 600 python bench.py --steps 20 --warmup 5 --no-end-to-end --no-config3 --no-config4 --no-config5 --no-in-flight --cpu-runs 1 2>/dev/null | tail -1 > gpurun_out/r04_mi2.json
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04_mi2.json").read())
fs = d["full_scoring"]
print("headline", round(d["ms_per_step"], 4), "full_scoring", round(fs["ms_per_pair"], 4), "mi", fs["stage_ms"].get("mutual_info"), "zncc", fs["stage_ms"].get("zncc"), "frac", round(fs["roofline"]["frac"], 3), "gate", fs["parity"]["passed"], fs["parity"]["mi_score"]["max_abs_diff"], fs["parity"]["mutual_info_score"]["max_abs_diff"])
PY
done
