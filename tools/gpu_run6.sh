#!/bin/bash
mkdir -p gpurun_out
B="--steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end --no-config3 --no-config4 --no-config5 --no-in-flight --no-full-scoring"
for rep in 1 2 3; do for dep in 1 2; do
  timeout 300 python bench.py $B --depth $dep > gpurun_out/r04_depth_${dep}_$rep.json 2>/dev/null
  python - $dep $rep <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/r04_depth_{sys.argv[1]}_{sys.argv[2]}.json").read().strip().splitlines()[-1])
print("depth", sys.argv[1], "ms_per_step", round(d["ms_per_step"], 4), "median", d["step_spread"]["median_ms"], "max", d["step_spread"]["max_ms"], "settle", d["settle"].get("post_gc_windows"), d["settle"].get("post_gc_last_window_ms_per_step"), "eig", d["stage_ms"].get("min_eigen"))
PY
done; done
for m in 0 1 0 1; do
  if [ $m = 1 ]; then export KARIOS_HIP_MI_SPREAD=1; else unset KARIOS_HIP_MI_SPREAD; fi
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end --no-config3 --no-config4 --no-config5 --no-in-flight > gpurun_out/r04_mi_$m.json 2>/dev/null
  python - $m <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/r04_mi_{sys.argv[1]}.json").read().strip().splitlines()[-1])
fs = d["full_scoring"]
print("MI spread", sys.argv[1], "full_scoring ms", round(fs["ms_per_pair"], 4), "mi stage", fs["stage_ms"].get("mutual_info"), "zncc", fs["stage_ms"].get("zncc"))
PY
done
unset KARIOS_HIP_MI_SPREAD
timeout 900 python -m pytest tests/test_gpu_rccl.py tests/test_gpu_matcher_mirror.py -q -x -m gpu 2>&1 | tail -4
