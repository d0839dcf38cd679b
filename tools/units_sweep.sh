export KARIOS_HIP_LIB=$PWD/karios_amd/libkarios_hip_dev.so
for r in 32 48 64 96 128 192; do echo "EIG3_ROWS=$r"; KARIOS_HIP_EIG3_ROWS=$r python tools/units_probe.py config4 3 2>&1 | grep "batched stage"; done
for r in 32 64 96 160; do echo "LAP_ROWS=$r"; KARIOS_HIP_LAP_ROWS=$r python tools/units_probe.py config4 3 2>&1 | grep "batched stage"; done
