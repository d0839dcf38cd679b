#!/bin/bash
# Round-end GPU run: the whole -m gpu suite, then tools/final_measure.sh (bench lines, rocprofv3 summaries, PMC passes).
#   gpurun --timeout 3600 -- "bash tools/round_end.sh <tag> <commit>"
TAG=${1:-r04}; COMMIT=${2:-unknown}
timeout 2400 python -m pytest tests -q -x -m gpu > gpurun_out/${TAG}_round_end_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/${TAG}_round_end_tests.log | cut -c1-200
timeout 900 python -m pytest tests -q -m perf > gpurun_out/${TAG}_round_end_perf.log 2>&1; echo "perf rc=$?"; tail -3 gpurun_out/${TAG}_round_end_perf.log | cut -c1-200
bash tools/final_measure.sh $TAG $COMMIT > gpurun_out/${TAG}_final_measure.log 2>&1; tail -30 gpurun_out/${TAG}_final_measure.log | cut -c1-200
