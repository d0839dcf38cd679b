#!/bin/bash
mkdir -p gpurun_out
timeout 900 python tools/exchange_probe.py 150 --json > gpurun_out/r04_exchange_probe.json 2> gpurun_out/r04_exchange_probe.err; echo "probe rc=$?"
python - <<'PY'
import json
ln = [l for l in open("gpurun_out/r04_exchange_probe.json") if l.startswith("{")]
d = json.loads(ln[-1])
print({k: d[k] for k in ("plain_ms_per_step", "exchange_ms_per_step", "plain_median_ms", "exchange_median_ms", "ratio_ms_per_step", "ratio_median", "rows_per_run")})
for k in ("plain", "exchange"):
    print(k, [(round(r["ms_per_step"], 4), round(r["median_submit_interval_ms"], 4)) for r in d["runs"][k]])
PY
timeout 600 python tools/exchange_probe.py 150 2>/dev/null | grep "ms per step"
