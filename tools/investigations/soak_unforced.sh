#!/bin/bash
# Unforced parity soak: 4 tile workers (default knobs, images up to 1500 px) + 2 aux workers (ZNCC / MI / phase / shift / DN filter) side by side.
T=${1:-900}; S=${2:-70000000}   # (env such as KARIOS_HIP_VERIFY_UPLOAD=1 is inherited by the workers)
mkdir -p gpurun_out
pids=()
for i in 0 1 2 3; do
  KARIOS_ORACLE_THREADS=3 timeout $((T + 180)) python tools/fuzz_parity.py --seed $((S + i * 1000000)) --cases 100000000 --max-size 1500 --budget-s "$T" > gpurun_out/r03_unforced_w$i.log 2>&1 &
  pids+=($!)
done
for i in 4 5; do
  KARIOS_ORACLE_THREADS=2 timeout $((T + 180)) python tools/fuzz_parity.py --what aux --seed $((S + i * 1000000)) --cases 100000000 --budget-s "$T" > gpurun_out/r03_unforced_w$i.log 2>&1 &
  pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait "$p" || rc=1; done
grep -h "FAIL\|fuzz_parity:\|paths taken" gpurun_out/r03_unforced_w*.log | cut -c1-300
exit $rc
