#!/bin/bash
# final-build soaks: unforced (4 tile + 2 aux workers) and forced-path
bash tools/soak_unforced.sh 600 91000000 > gpurun_out/r04_final_unforced.log 2>&1; echo "unforced rc=$?"; tail -8 gpurun_out/r04_final_unforced.log | cut -c1-250
bash tools/fuzz_soak.sh 6 420 92000000 300 700 > gpurun_out/r04_final_forced.log 2>&1; echo "forced rc=$?"; tail -8 gpurun_out/r04_final_forced.log | cut -c1-250
