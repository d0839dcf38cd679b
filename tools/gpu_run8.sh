#!/bin/bash
mkdir -p gpurun_out
timeout 900 python tools/exchange_probe.py 150 --json 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1])
print('lagged', {k: round(d[k],4) if isinstance(d[k],float) else d[k] for k in ('plain_ms_per_step','exchange_ms_per_step','ratio_ms_per_step','ratio_median')})
for k in ('plain','exchange'): print(k, [(round(r['ms_per_step'],4)) for r in d['runs'][k]])"
timeout 900 python tools/exchange_probe.py 150 --json --at-submit 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1])
print('at submit', {k: round(d[k],4) if isinstance(d[k],float) else d[k] for k in ('plain_ms_per_step','exchange_ms_per_step','ratio_ms_per_step','ratio_median')})
for k in ('plain','exchange'): print(k, [(round(r['ms_per_step'],4)) for r in d['runs'][k]])"
B="--steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end --no-config3 --no-config4 --no-config5 --no-in-flight --no-full-scoring"
for q in 4 8 4 8; do
  GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py $B > gpurun_out/r04_q$q.json 2>/dev/null
  python - $q <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/r04_q{sys.argv[1]}.json").read().strip().splitlines()[-1])
print("GPU_MAX_HW_QUEUES", sys.argv[1], "ms_per_step", round(d["ms_per_step"], 4), "median", d["step_spread"]["median_ms"], "lk", d["stage_ms"].get("lk_fwd_bwd"), "minmax", d["stage_ms"].get("minmax"), "pyr", d["stage_ms"].get("pyramid"))
PY
done
KARIOS_BENCH_EXCHANGE=1 timeout 300 python bench.py $B > gpurun_out/r04_ex.json 2>/dev/null; python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04_ex.json").read().strip().splitlines()[-1])
print("bench with exchange", round(d["ms_per_step"], 4), d["step_spread"]["median_ms"], json.dumps(d["exchange"])[:400])
PY
timeout 900 python -m pytest tests/test_gpu_rccl.py -q -x -m gpu 2>&1 | tail -12
