#!/bin/bash
# usage: tools/pmc_run.sh <tag> <counters...>   (run on the GPU box; one --pmc pass, kernel-trace only)
R=$PWD; tag=$1; shift
mkdir -p $R/gpurun_out/pmc_$tag
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_$tag -o $tag -- python3 $R/tools/step_breakdown.py > $R/gpurun_out/pmc_$tag/run.log 2>&1
cd $R
python3 tools/pmc_summary.py gpurun_out/pmc_$tag/${tag}_counter_collection.csv
