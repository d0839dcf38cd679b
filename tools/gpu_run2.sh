#!/bin/bash
# round 4, GPU call 2: one-rank RCCL tests, MI-fused frames, a bench line with full_scoring, MI kernel A/B, ring bandwidth, the two soaks
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_rccl.py tests/test_gpu_matcher_mirror.py tests/test_gpu_golden.py -q -x -m gpu > gpurun_out/r04_run2_tests.log 2>&1; echo "tests rc=$?"
tail -15 gpurun_out/r04_run2_tests.log
for f in 1 0; do
  if [ $f = 1 ]; then export KARIOS_HIP_MI_FIRST_FORM=1; else unset KARIOS_HIP_MI_FIRST_FORM; fi
  echo "MI first form = $f"; timeout 300 python tools/entry_points_workload.py 10980 scoring 2>&1 | tail -2
done
for t in 4 8 12 16; do KARIOS_HIP_COPY_THREADS=$t timeout 200 python tools/ring_bw.py 2>&1 | tail -2; done
timeout 900 python bench.py --steps 20 --warmup 5 --no-config3 --no-config5 --no-in-flight --cpu-runs 1 > gpurun_out/r04_bench_b.json 2> gpurun_out/r04_bench_b.err; echo "bench rc=$?"
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r04_bench_b.json").read().strip().splitlines()[-1])
    print("ms_per_step", d["ms_per_step"], "parity", d["cpu_baseline"]["parity"].get("passed"))
    fs = d.get("full_scoring", {})
    print("full_scoring", {k: fs.get(k) for k in ("ms_per_pair", "scored_rows_per_pair", "stage_ms")}, "roof", fs.get("roofline", {}).get("frac"), "gate", json.dumps(fs.get("parity"))[:400])
    print("end_to_end pageable", d.get("end_to_end", {}).get("pageable_numpy_ms_per_pair"), "pinned", d.get("end_to_end", {}).get("ms_per_pair"))
    print("config4 ab", json.dumps(d.get("config4", {}).get("contexts_in_flight_ab")))
except Exception as e:
    print("bench parse failed", e); print(open("gpurun_out/r04_bench_b.err").read()[-2000:])
PY
KARIOS_HIP_ASYNC_HOST_UPLOAD=1 KARIOS_HIP_UPLOAD_CHECKSUM=1 bash tools/soak_unforced.sh 540 91000000 > gpurun_out/r04_soak_old.log 2>&1; echo "old soak rc=$?"
mkdir -p gpurun_out/r04_soak_old_logs; cp gpurun_out/r03_unforced_w*.log gpurun_out/r04_soak_old_logs/ 2>/dev/null
tail -12 gpurun_out/r04_soak_old.log; grep -h "UPLOAD_CHECKSUM" gpurun_out/r03_unforced_w*.log | head -5
KARIOS_HIP_UPLOAD_CHECKSUM=1 bash tools/soak_unforced.sh 420 95000000 > gpurun_out/r04_soak_ring.log 2>&1; echo "ring soak rc=$?"
mkdir -p gpurun_out/r04_soak_ring_logs; cp gpurun_out/r03_unforced_w*.log gpurun_out/r04_soak_ring_logs/ 2>/dev/null
tail -12 gpurun_out/r04_soak_ring.log; grep -h "UPLOAD_CHECKSUM" gpurun_out/r03_unforced_w*.log | head -5
