#!/bin/bash
# The round's closing sweeps at the final library: tile (plain and forced paths), batched units, aux entry points, pipelined streams -
# side by side on the one GPU, T seconds each.  -> gpurun_out/final_soak_*.log
T=${1:-240}; S=${2:-7000001}
mkdir -p gpurun_out
KARIOS_ORACLE_THREADS=3 python tools/fuzz_parity.py --what tile --seed $S --cases 100000000 --max-size 900 --budget-s $T > gpurun_out/final_soak_tile.log 2>&1 &
KARIOS_ORACLE_THREADS=3 python tools/fuzz_parity.py --what tile --force-paths --seed $((S + 100000)) --cases 100000000 --max-size 1100 --budget-s $T > gpurun_out/final_soak_forced.log 2>&1 &
KARIOS_ORACLE_THREADS=3 python tools/fuzz_parity.py --what units --seed $((S + 200000)) --cases 100000000 --budget-s $T > gpurun_out/final_soak_units.log 2>&1 &
KARIOS_ORACLE_THREADS=3 python tools/fuzz_parity.py --what aux --seed $((S + 300000)) --cases 100000000 --budget-s $T > gpurun_out/final_soak_aux.log 2>&1 &
python tools/fuzz_pipeline.py --rounds 100000 --seed $((S + 400000)) --budget-s $T > gpurun_out/final_soak_pipeline.log 2>&1 &
wait
grep -h "FAIL\|fuzz_parity:\|fuzz_pipeline:\|paths taken\|units mode" gpurun_out/final_soak_*.log | cut -c1-300
