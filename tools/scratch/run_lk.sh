set -x
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pyrlk or lk_forward or klt_track or klt_tile" 2>&1 | tail -8
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-end-to-end --no-config3 --no-config4 --no-config5 --no-in-flight"
$B > gpurun_out/lk2_on.json 2>/dev/null
KARIOS_HIP_OPTIONS=lk2=0 $B > gpurun_out/lk2_off.json 2>/dev/null
python - <<'PY'
import json
for f in ("lk2_on","lk2_off"):
    d=json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
    print(f, d["ms_per_step"], d["stage_ms"])
PY
