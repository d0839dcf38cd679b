#!/bin/bash
# usage: minmax_race.sh <seconds>   4 checkers + 2 load generators (aux fuzz: FFT / ZNCC / MI kernels) side by side
T=${1:-200}
pids=()
for i in 4 5; do KARIOS_ORACLE_THREADS=2 timeout $((T + 120)) python tools/fuzz_parity.py --what aux --seed $((99000000 + i * 1000000)) --cases 100000000 --budget-s "$T" > /dev/null 2>&1 & pids+=($!); done
for i in 0 1 2 3; do timeout $((T + 120)) python tools/scratch/minmax_race.py "$T" $((i + 1)) 2>&1 | grep -v amdgpu.ids & pids+=($!); done
for p in "${pids[@]}"; do wait "$p"; done
