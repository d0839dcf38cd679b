#!/bin/bash
timeout 2400 python -m pytest tests -q -x -m gpu > gpurun_out/r04_run25_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r04_run25_tests.log | cut -c1-200
bash tools/final_measure.sh r04 $1 > gpurun_out/r04_final_measure.log 2>&1; tail -60 gpurun_out/r04_final_measure.log | cut -c1-220
