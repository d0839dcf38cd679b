#!/bin/bash
# Round-end measurements on the GPU box: the driver's bench command, the config-3 line, and the rocprofv3 kernel statistics of the
# headline loop / of config 3; results under gpurun_out/ (copied to profiles/ by hand).
R=$PWD; TAG=${1:-r03}
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_${TAG}.json 2> gpurun_out/bench_${TAG}.err
python bench.py > gpurun_out/bench_${TAG}_default.json 2> gpurun_out/bench_${TAG}_default.err
python bench.py --config 3 --steps 10 --warmup 2 > gpurun_out/bench_${TAG}_config3.json 2> gpurun_out/bench_${TAG}_config3.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_c2 -o c2 -- python3 $R/bench.py --no-cpu-baseline --no-end-to-end --no-in-flight --no-config3 --no-config4 --no-config5 --steps 20 --warmup 5 > $R/gpurun_out/prof_${TAG}_c2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_c3 -o c3 -- python3 $R/bench.py --config 3 --steps 5 --warmup 1 > $R/gpurun_out/prof_${TAG}_c3.log 2>&1
cd $R
python3 tools/summarize_rocprof.py $(find gpurun_out/prof_${TAG}_c2 -name 'c2_kernel_stats.csv' | head -1) gpurun_out/${TAG}_kernel_stats_config2.md
python3 tools/summarize_rocprof.py $(find gpurun_out/prof_${TAG}_c3 -name 'c3_kernel_stats.csv' | head -1) gpurun_out/${TAG}_kernel_stats_config3.md 6
tail -1 gpurun_out/bench_${TAG}.json | cut -c1-300
head -24 gpurun_out/${TAG}_kernel_stats_config2.md
head -14 gpurun_out/${TAG}_kernel_stats_config3.md
