#!/bin/bash
mkdir -p gpurun_out
timeout 900 python tools/exchange_probe.py 150 --json 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1])
print('batched', {k: round(d[k],4) if isinstance(d[k],float) else d[k] for k in ('plain_ms_per_step','exchange_ms_per_step','ratio_ms_per_step','ratio_median')})
for k in ('plain','exchange'): print(k, [(round(r['ms_per_step'],4)) for r in d['runs'][k]])"
B="--steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end --no-config3 --no-config4 --no-config5 --no-in-flight --no-full-scoring"
for e in 0 1 0 1; do
  if [ $e = 1 ]; then export KARIOS_BENCH_EXCHANGE=1; else unset KARIOS_BENCH_EXCHANGE; fi
  timeout 300 python bench.py $B 2>/dev/null | tail -1 > gpurun_out/r04_ex_$e.json
  python - $e <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/r04_ex_{sys.argv[1]}.json").read())
print("exchange", sys.argv[1], "ms_per_step", round(d["ms_per_step"], 4), "median", d["step_spread"]["median_ms"], "eig", d["stage_ms"]["min_eigen"], (d["exchange"] or {}).get("rows_from_gathered_blocks"))
PY
done
unset KARIOS_BENCH_EXCHANGE
timeout 900 python -m pytest tests/test_gpu_rccl.py tests/test_gpu_bench.py -q -x -m gpu 2>&1 | tail -6
