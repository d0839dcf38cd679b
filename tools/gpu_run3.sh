#!/bin/bash
# round 4, GPU call 3: LK pair form (parity + A/B), exchange cost breakdown, ring bandwidth sweep
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_forced_paths.py tests/test_gpu_fuzz.py tests/test_gpu_golden.py tests/test_gpu_matcher_mirror.py -q -x -m gpu > gpurun_out/r04_run3_tests.log 2>&1; echo "tests rc=$?"
tail -6 gpurun_out/r04_run3_tests.log
B="--steps 60 --warmup 10 --no-cpu-baseline --no-end-to-end --no-config3 --no-config4 --no-config5 --no-in-flight --no-full-scoring"
for o in "lk_pair=1" "lk_pair=0" "lk_pair=1" "lk_pair=0"; do
  KARIOS_HIP_OPTIONS="$o" timeout 300 python bench.py $B > gpurun_out/r04_ab_$o.json 2>/dev/null
  python - "$o" <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/r04_ab_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print(sys.argv[1], "ms_per_step", round(d["ms_per_step"], 4), "median", d["step_spread"]["median_ms"], "lk", d["stage_ms"].get("lk_fwd_bwd"), "zncc", d["stage_ms"].get("zncc"), "kp", d["matched_keypoints_per_pair"])
PY
done
timeout 600 python tools/exchange_probe.py 200 2>&1 | grep "ms per step"
for t in 2 3 4 5 6; do KARIOS_HIP_COPY_THREADS=$t timeout 200 python tools/ring_bw.py 2>&1 | grep pageable; done
for k in 1024 2048 8192; do KARIOS_HIP_RING_CHUNK_KB=$k timeout 200 python tools/ring_bw.py 2>&1 | grep pageable | sed "s/^/chunk ${k} KB: /"; done
timeout 900 python -m pytest tests/test_gpu_rccl.py -q -x -m gpu 2>&1 | tail -8
