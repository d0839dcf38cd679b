#!/bin/bash
R=$PWD
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_WAIT_ANY"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  mkdir -p $R/gpurun_out/pmc_c3/$tag
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc_c3/$tag -o p -- python3 $R/bench.py --config 3 --steps 2 --warmup 1 > $R/gpurun_out/pmc_c3/$tag/run.log 2>&1
  cd $R
  python3 tools/pmc_summary.py gpurun_out/pmc_c3/$tag/p_counter_collection.csv | grep fft_rows
done
