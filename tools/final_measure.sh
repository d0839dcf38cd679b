#!/bin/bash
# Round-end measurements on the GPU box: bench lines (config 2 incl. end_to_end / config4 / cpu_baseline, config 3) and the
# rocprofv3 kernel statistics of the same bench command; results under gpurun_out/ (copied to profiles/ by hand).
R=$PWD
python bench.py > gpurun_out/bench_r02_final.json 2> gpurun_out/bench_r02_final.err
python bench.py --config 3 > gpurun_out/bench_r02_config3.json 2> gpurun_out/bench_r02_config3.err
python bench.py --in-flight --no-cpu-baseline --no-end-to-end --no-config4 > gpurun_out/bench_r02_in_flight.json 2> gpurun_out/bench_r02_in_flight.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r02_c2 -o c2 -- python3 $R/bench.py --no-cpu-baseline --no-end-to-end --no-config4 --steps 20 --warmup 5 > $R/gpurun_out/prof_r02_c2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r02_c3 -o c3 -- python3 $R/bench.py --config 3 --steps 5 --warmup 1 > $R/gpurun_out/prof_r02_c3.log 2>&1
cd $R
python3 tools/summarize_rocprof.py $(find gpurun_out/prof_r02_c2 -name 'c2_kernel_stats.csv' | head -1) gpurun_out/r02_kernel_stats_config2.md 37
python3 tools/summarize_rocprof.py $(find gpurun_out/prof_r02_c3 -name 'c3_kernel_stats.csv' | head -1) gpurun_out/r02_kernel_stats_config3.md 6
tail -1 gpurun_out/bench_r02_final.json | cut -c1-400
tail -1 gpurun_out/bench_r02_config3.json | cut -c1-600
head -20 gpurun_out/r02_kernel_stats_config2.md
