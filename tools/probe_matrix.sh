echo "--- release"; python tools/two_contexts_probe.py 1 2 2>&1 | grep context
for r in 32 48 64; do
echo "--- dev rows=$r"; KARIOS_HIP_LIB=$PWD/karios_amd/libkarios_hip_dev.so KARIOS_HIP_EIG3_ROWS=$r python tools/two_contexts_probe.py 1 2>&1 | grep context
done
