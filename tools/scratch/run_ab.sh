# usage: run_ab.sh "<opts A>" "<opts B>" ...   each a KARIOS_HIP_OPTIONS string ("-" = none); prints ms/step + stage table
B="python bench.py $BENCH_EXTRA --steps 40 --warmup 5 --no-cpu-baseline --no-end-to-end --no-config3 --no-config4 --no-config5 --no-in-flight"
i=0
for o in "$@"; do
  i=$((i+1))
  if [ "$o" = "-" ]; then $B > gpurun_out/ab_$i.json 2>/dev/null; else KARIOS_HIP_OPTIONS="$o" $B > gpurun_out/ab_$i.json 2>/dev/null; fi
  python - "$o" gpurun_out/ab_$i.json <<'PY'
import json, sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
s=d["stage_ms"]
print(f"{sys.argv[1]:28s} ms/step {d['ms_per_step']:.4f} settle {d['settle']['last_window_ms_per_step']} | " + " ".join(f"{k[:6]}={v:.3f}" for k,v in s.items() if v>0), "| kp", d["matched_keypoints_per_pair"], d["median_dx_dy"], "redone", d["speculative_tiles_redone"])
PY
done
