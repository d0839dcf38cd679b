#!/bin/bash
mkdir -p gpurun_out
timeout 300 python tools/rccl_banner_probe.py 2>&1 | tail -6
timeout 2400 python -m pytest tests -q -x -m gpu --durations=12 > gpurun_out/r04_run17_tests.log 2>&1; echo "tests rc=$?"; tail -22 gpurun_out/r04_run17_tests.log | cut -c1-200
