#!/bin/bash
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r02f -o f -- python3 $R/bench.py --no-cpu-baseline --no-end-to-end --no-config4 --steps 12 --warmup 3 > $R/gpurun_out/prof_r02f.log 2>&1
cd $R
f=$(find gpurun_out/prof_r02f -name 'f_kernel_trace.csv' | head -1)
echo $f
python3 tools/trace_gaps.py $f -v | tail -75
