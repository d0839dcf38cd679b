#!/bin/bash
# eig3 tuning: stage time of the fused eig+candidate pass for several item heights, against the 2-px kernel
show() { python - "$1" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
s=d["stage_ms"]; print(f'{d["ms_per_step"]:.3f} ms/step  eig {s.get("min_eigen_candidates_fused",0):.4f}  sort {s["sort"]:.3f} select {s["select"]:.3f} lk {s["lk_fwd_bwd"]:.3f} n_cand {d["n_candidates"]} kp {d["matched_keypoints_per_pair"]} med {d["median_dx_dy"]}')
PY
}
for cfg in "KARIOS_HIP_EIG3=0" "KARIOS_HIP_EIG3=1" "KARIOS_HIP_EIG3_ROWS=32" "KARIOS_HIP_EIG3_ROWS=48" "KARIOS_HIP_EIG3_ROWS=64" "KARIOS_HIP_EIG3_ROWS=96" "KARIOS_HIP_EIG3_ROWS=128" "KARIOS_HIP_EIG3_ROWS=192" "KARIOS_HIP_SPECULATIVE=1"; do
  env $cfg python bench.py --no-cpu-baseline --no-end-to-end --no-config4 --steps 20 > gpurun_out/tune.json 2> gpurun_out/tune.err || tail -3 gpurun_out/tune.err
  echo -n "$cfg: "; show gpurun_out/tune.json
done
