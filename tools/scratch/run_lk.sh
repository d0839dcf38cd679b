python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pyrlk or lk_ or klt_track or klt_tile" 2>&1 | tail -3
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-end-to-end --no-config3 --no-config4 --no-config5 --no-in-flight"
$B > gpurun_out/v_w6_on.json 2>/dev/null
KARIOS_HIP_OPTIONS=lk_order=0 $B > gpurun_out/v_w6_off.json 2>/dev/null
KARIOS_HIP_LIB=$PWD/karios_amd/libkarios_hip_w0.so KARIOS_HIP_OPTIONS=lk_order=0 $B > gpurun_out/v_w0_off.json 2>/dev/null
KARIOS_HIP_LIB=$PWD/karios_amd/libkarios_hip_w7.so KARIOS_HIP_OPTIONS=lk_order=0 $B > gpurun_out/v_w7_off.json 2>/dev/null
KARIOS_HIP_LIB=$PWD/karios_amd/libkarios_hip_w7.so $B > gpurun_out/v_w7_on.json 2>/dev/null
python - <<'PY'
import json
for f in ("v_w6_on","v_w6_off","v_w0_off","v_w7_off","v_w7_on"):
    d=json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
    print(f, round(d["ms_per_step"],4), d["stage_ms"]["lk_fwd_bwd"], d["median_dx_dy"], d["matched_keypoints_per_pair"])
PY
