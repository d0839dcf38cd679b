#!/bin/bash
KARIOS_HIP_UPLOAD_CHECKSUM=1 bash tools/soak_unforced.sh 1200 181000000 > gpurun_out/r04_final4_unforced_chk.log 2>&1; echo "unforced(checksum armed) rc=$?"; grep -h "fuzz_parity:\|MISS" gpurun_out/r04_final4_unforced_chk.log | cut -c1-160; grep -c "MISS" gpurun_out/r03_unforced_w*.log
