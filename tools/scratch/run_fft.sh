python -m pytest tests/test_gpu_matcher_mirror.py tests/test_gpu_golden.py tests/test_gpu_parity.py -x -q -m gpu -k "phase or large_offset" 2>&1 | tail -5
show='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["ms_per_step"],3), d["stage_ms"], d["detected_offset_row_col"], d["phase_path"], d.get("gate",{}).get("passed"))'
python bench.py --config 3 --steps 6 --warmup 2 2>/dev/null | python -c "$show"
KARIOS_HIP_OPTIONS=fft61=0 python bench.py --config 3 --steps 6 --warmup 2 2>/dev/null | python -c "$show"
