B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-config3 --no-config4 --no-config5"
$B > gpurun_out/q_e2e.json 2>/dev/null
$B --no-end-to-end > gpurun_out/q_noe2e.json 2>/dev/null
python - <<'PY'
import json
for f in ("q_e2e","q_noe2e"):
    d=json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
    print(f, round(d["ms_per_step"],4), d["in_flight"]["ms_per_pair"], d.get("end_to_end",{}).get("ms_per_pair"))
PY
