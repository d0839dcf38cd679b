#!/bin/bash
# usage (on the GPU box): PMC_COMMIT=<sha> tools/pmc_collect.sh <tag> <python script> [args ...]
# Three separate rocprofv3 --pmc passes (kernel trace only, the program itself after "--") over the same workload: FETCH_SIZE,
# WRITE_SIZE, instruction / activity counters.  Results land in gpurun_out/pmc_<tag>/{f,w,v}; tools/pmc_to_json.py folds them
# into profiles/pmc_traffic.json.
R=$PWD; tag=$1; shift
out=$R/gpurun_out/pmc_$tag
mkdir -p $out
echo "${PMC_COMMIT:-unknown}" > $out/commit.txt
script=$R/$1; shift
cd /tmp && export TMPDIR=/tmp
run_pass() {
  name=$1; shift
  mkdir -p $out/$name
  timeout 280 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out/$name -o p -- python3 $script $ARGS > $out/$name/run.log 2>&1
}
ARGS="$*"
run_pass f FETCH_SIZE
run_pass w WRITE_SIZE
run_pass v SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE
cd $R
