echo "--- release, no hp"; python tools/two_contexts_probe.py 1 2 3 2>&1 | grep context
echo "--- release, hp"; KARIOS_PROBE_HP=1 python tools/two_contexts_probe.py 1 2 2>&1 | grep context
for r in 48 64; do
echo "--- dev rows=$r no hp"; KARIOS_HIP_LIB=$PWD/karios_amd/libkarios_hip_dev.so KARIOS_HIP_EIG3_ROWS=$r python tools/two_contexts_probe.py 1 2 2>&1 | grep context
done
