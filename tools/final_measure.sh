#!/bin/bash
# Round-end measurements on the GPU box: the driver's bench command, the default line, the config-3 line, the rocprofv3 kernel
# statistics of the headline loop (+ full scoring), of config 3 and of the entry points the loop does not reach, and the PMC passes
# (separate --pmc runs, kernel trace only).  Results under gpurun_out/ (the summaries are copied to profiles/ by hand).
R=$PWD; TAG=${1:-r04}; COMMIT=${2:-unknown}
python bench.py --gpus 1 --steps 20 --warmup 5 2> gpurun_out/bench_${TAG}.err | tail -1 > gpurun_out/bench_${TAG}.json
python bench.py 2> gpurun_out/bench_${TAG}_default.err | tail -1 > gpurun_out/bench_${TAG}_default.json
python bench.py --config 3 --steps 10 --warmup 2 2> gpurun_out/bench_${TAG}_config3.err | tail -1 > gpurun_out/bench_${TAG}_config3.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_c2 -o c2 -- python3 $R/bench.py --no-cpu-baseline --no-end-to-end --no-in-flight --no-config3 --no-config4 --no-config5 --steps 20 --warmup 5 > $R/gpurun_out/prof_${TAG}_c2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_c3 -o c3 -- python3 $R/bench.py --config 3 --steps 5 --warmup 1 > $R/gpurun_out/prof_${TAG}_c3.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_ep -o ep -- python3 $R/tools/entry_points_workload.py 10980 scoring,dn,auto,banded > $R/gpurun_out/prof_${TAG}_ep.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_f64 -o f64 -- python3 $R/tools/phase64_workload.py 10980 5 > $R/gpurun_out/prof_${TAG}_f64.log 2>&1
cd $R
python3 tools/summarize_rocprof.py $(find gpurun_out/prof_${TAG}_f64 -name 'f64_kernel_stats.csv' | head -1) gpurun_out/${TAG}_kernel_stats_phase_fp64.md 5
python3 tools/summarize_rocprof.py $(find gpurun_out/prof_${TAG}_c2 -name 'c2_kernel_stats.csv' | head -1) gpurun_out/${TAG}_kernel_stats_config2.md
python3 tools/summarize_rocprof.py $(find gpurun_out/prof_${TAG}_c3 -name 'c3_kernel_stats.csv' | head -1) gpurun_out/${TAG}_kernel_stats_config3.md 6
python3 tools/summarize_rocprof.py $(find gpurun_out/prof_${TAG}_ep -name 'ep_kernel_stats.csv' | head -1) gpurun_out/${TAG}_kernel_stats_entry_points.md
PMC_COMMIT=$COMMIT bash tools/pmc_collect.sh ${TAG}_c2 bench.py --no-cpu-baseline --no-end-to-end --no-in-flight --no-config3 --no-config4 --no-config5 --no-full-scoring --steps 12 --warmup 3
PMC_COMMIT=$COMMIT bash tools/pmc_collect.sh ${TAG}_ep tools/entry_points_workload.py 10980 scoring,dn
PMC_COMMIT=$COMMIT bash tools/pmc_collect.sh ${TAG}_c3 bench.py --config 3 --steps 4 --warmup 1
PMC_COMMIT=$COMMIT bash tools/pmc_collect.sh ${TAG}_f64 tools/phase64_workload.py 10980 3
tail -1 gpurun_out/bench_${TAG}.json | cut -c1-300
head -30 gpurun_out/${TAG}_kernel_stats_config2.md
head -14 gpurun_out/${TAG}_kernel_stats_config3.md
head -30 gpurun_out/${TAG}_kernel_stats_entry_points.md
head -16 gpurun_out/${TAG}_kernel_stats_phase_fp64.md
