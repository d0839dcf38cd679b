#!/bin/bash
# round 4, GPU call 4: fused FFT transposes (parity + A/B), MI-fused frames, exchange in-process cadence, ring default
mkdir -p gpurun_out
timeout 1800 python -m pytest tests/test_gpu_matcher_mirror.py tests/test_gpu_rccl.py -q -x -m gpu > gpurun_out/r04_run4_tests.log 2>&1; echo "tests rc=$?"
tail -8 gpurun_out/r04_run4_tests.log
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -x -m gpu -k "config3" 2>&1 | tail -3
for o in "fft_ts=1" "fft_ts=0" "fft_ts=1" "fft_ts=0"; do
  KARIOS_HIP_OPTIONS="$o" timeout 400 python bench.py --config 3 --steps 10 --warmup 2 > gpurun_out/r04_c3_$o.json 2>/dev/null
  python - "$o" <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/r04_c3_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print(sys.argv[1], "ms_per_step", round(d["ms_per_step"], 4), "phase", d.get("stage_ms", {}).get("phase_correlation"), "roof", d.get("roofline", {}).get("frac"), "off", d.get("detected_offset_row_col"), "gate", d.get("gate", {}).get("passed"))
PY
done
timeout 200 python tools/ring_bw.py 2>&1 | grep -v amdgpu.ids
timeout 900 python bench.py --steps 20 --warmup 5 --no-config3 --no-config5 --no-in-flight --no-cpu-baseline --no-full-scoring > gpurun_out/r04_bench_c.json 2> gpurun_out/r04_bench_c.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04_bench_c.json").read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], "e2e", d["end_to_end"]["ms_per_pair"], "pageable", d["end_to_end"]["pageable_numpy_ms_per_pair"])
print("config4", d["config4"]["ms_per_step"], json.dumps(d["config4"]["contexts_in_flight_ab"])[:300])
PY
