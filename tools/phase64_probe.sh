#!/bin/bash
timeout 1500 python tools/phase64_probe.py > gpurun_out/r04_phase64_probe.log 2>&1; echo "probe rc=$?"; tail -40 gpurun_out/r04_phase64_probe.log | cut -c1-230
