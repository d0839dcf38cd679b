#!/bin/bash
# Copy what tools/round_end.sh left under gpurun_out/ into profiles/ (run HERE, after the gpurun call) and fold the PMC passes.
#   bash tools/collect_record.sh <tag>
TAG=${1:-r06}
for n in bench_$TAG.json bench_${TAG}_detail.json bench_${TAG}_default.json bench_${TAG}_default_detail.json bench_${TAG}_config3.json bench_${TAG}_config3_detail.json; do cp gpurun_out/$n profiles/$n; done
cp gpurun_out/bench_$TAG.out profiles/bench_${TAG}_stdout.txt
for f in config2 config2_driver_flags config2_one_pair_per_submission config3 config4_batched config4_pipelined_loop e2e_shape_batched entry_points kernels_alone phase_fp64; do
  cp gpurun_out/${TAG}_kernel_stats_$f.md profiles/
done
cp gpurun_out/timeline_${TAG}_headline.txt profiles/${TAG}_timeline_headline.txt
cp gpurun_out/timeline_${TAG}_c4.txt profiles/${TAG}_timeline_config4.txt
python tools/pmc_to_json.py gpurun_out/pmc_${TAG}_c2 10980 --mode=config2_batched --units-per-launch=4 | tail -10
python tools/pmc_to_json.py gpurun_out/pmc_${TAG}_ep 10980 --mode=scoring --only=mi_kernel | tail -1
python tools/pmc_to_json.py gpurun_out/pmc_${TAG}_ep 10980 --mode=dn --only=dn_keep | tail -1
python tools/pmc_to_json.py gpurun_out/pmc_${TAG}_c3 10980 --config3 --only=phase_correlation_f32,shift_image | tail -2
python tools/pmc_to_json.py gpurun_out/pmc_${TAG}_f64 10980 --mode=f64 --only=phase_correlation_f64,phase_f64_prime_level,phase_f64_smooth_level,phase_f64_cross_power | tail -4
python - <<P
import json
for n in ("bench_$TAG", "bench_${TAG}_default", "bench_${TAG}_config3"):
    raw = open(f"profiles/{n}.json").read()
    d = json.loads(raw)
    print(n, len(raw), "bytes:", round(d["value"], 1), d["unit"], round(d["ms_per_step"], 4), "ms per step, gates", d.get("gates_all_passed"))
P
