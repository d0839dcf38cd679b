#!/bin/bash
# Forced-path parity soak: N fuzz_parity.py workers side by side on the one GPU (different seed ranges, two oracle threads
# each), every case with the km_set_option test knobs drawn so that the corner detector's retry paths run.
#   tools/fuzz_soak.sh <workers> <seconds> <first seed> [max size] [max size of the last three workers (wide images: 8-px eig kernel)]
# Per-worker summaries -> gpurun_out/r03_soak_w<i>.log; any failing case dumps its arrays (gpurun_out/fuzz_fail_*.npz).
N=${1:-6}; T=${2:-600}; S=${3:-1000000}; M=${4:-300}; M2=${5:-$M}
mkdir -p gpurun_out
pids=()
for i in $(seq 0 $((N - 1))); do
  KARIOS_ORACLE_THREADS=2 timeout $((T + 120)) python tools/fuzz_parity.py --force-paths --seed $((S + i * 10000000)) --cases 100000000 \
      --max-size "$([ $i -ge $((N - 3)) ] && echo $M2 || echo $M)" --budget-s "$T" > gpurun_out/r03_soak_w$i.log 2>&1 &
  pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait "$p" || rc=1; done
grep -h "FAIL\|fuzz_parity:\|paths taken" gpurun_out/r03_soak_w*.log | cut -c1-300
exit $rc
