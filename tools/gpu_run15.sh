#!/bin/bash
# final-build soaks: forced retry paths (6 workers) and the unforced large-image mix, detector armed
mkdir -p gpurun_out
bash tools/fuzz_soak.sh 6 420 110000000 300 700 > gpurun_out/r04_soak_forced.log 2>&1; echo "forced rc=$?"; tail -8 gpurun_out/r04_soak_forced.log | cut -c1-200
KARIOS_HIP_UPLOAD_CHECKSUM=1 bash tools/soak_unforced.sh 360 120000000 > gpurun_out/r04_soak_final.log 2>&1; echo "unforced rc=$?"; tail -8 gpurun_out/r04_soak_final.log | cut -c1-200
grep -h "UPLOAD_CHECKSUM" gpurun_out/r03_unforced_w*.log gpurun_out/r03_soak_w*.log | head -3
