#!/bin/bash
bash tools/soak_unforced.sh 540 151000000 > gpurun_out/r04_final2_unforced.log 2>&1; echo "unforced rc=$?"; grep -h "fuzz_parity:" gpurun_out/r04_final2_unforced.log | cut -c1-120
