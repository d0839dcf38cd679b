# usage: env_ab.sh VAR v1 v2 ...   runs the short bench with VAR=v for each value ("-" = unset); prints ms/step + stage table
VAR=$1; shift
B="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-end-to-end --no-config3 --no-config4 --no-config5 --no-in-flight"
i=0
for v in "$@"; do
  i=$((i+1))
  if [ "$v" = "-" ]; then $B > gpurun_out/eab_$i.json 2>/dev/null; else env $VAR=$v $B > gpurun_out/eab_$i.json 2>/dev/null; fi
  python - "$VAR=$v" gpurun_out/eab_$i.json <<'PY'
import json, sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
s=d["stage_ms"]
print(f"{sys.argv[1]:28s} ms/step {d['ms_per_step']:.4f} settle {d['settle']['last_window_ms_per_step']} | " + " ".join(f"{k[:6]}={v:.3f}" for k,v in s.items() if v>0), "| kp", d["matched_keypoints_per_pair"], "redone", d["speculative_tiles_redone"])
PY
done
