#!/bin/bash
R=$PWD
for q in 4 8; do
for ov in 1 0; do
GPU_MAX_HW_QUEUES=$q KARIOS_HIP_TAIL_OVERLAP=$ov python bench.py --no-cpu-baseline --no-end-to-end --no-in-flight --no-config3 --no-config4 --no-config5 --no-full-scoring --steps 60 --warmup 10 2>/dev/null | tail -1 > gpurun_out/r04_tail_$ov.json
python3 - <<PY
import json
d=json.load(open('gpurun_out/r04_tail_$ov.json'))
print('queues', $q, 'overlap', $ov, 'ms_per_step', round(d['ms_per_step'],4))
PY
done
done
cd /tmp && export TMPDIR=/tmp
KARIOS_HIP_TAIL_OVERLAP=1 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_tail -o t -- python3 $R/bench.py --no-cpu-baseline --no-end-to-end --no-in-flight --no-config3 --no-config4 --no-config5 --no-full-scoring --steps 12 --warmup 3 > $R/gpurun_out/prof_tail.log 2>&1
cd $R
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof_tail/**/t_kernel_trace.csv', recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last 40 kernels before the final lk2 launch
idx=[i for i,r in enumerate(rows) if 'lk2_kernel' in r['Kernel_Name']]
i0=idx[-3]; t0=int(rows[i0]['Start_Timestamp'])
for r in rows[i0:idx[-2]+6]:
    print(f"{(int(r['Start_Timestamp'])-t0)/1e3:9.1f} {(int(r['End_Timestamp'])-t0)/1e3:9.1f} q{r.get('Queue_Id','?')} s{r.get('Stream_Id','?')} {r['Kernel_Name'][:50]}")
PY
