#!/bin/bash
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_f64 -o f64 -- python3 $R/tools/phase64_workload.py 10980 5 ${DBG:-0} > $R/gpurun_out/prof_f64.log 2>&1
cd $R
tail -1 gpurun_out/prof_f64.log
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_f64/f64_kernel_trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
ks=[r for r in rows if 'f64' in r['Kernel_Name']]
n=len(ks)//5
tot=0
for r in ks[-n:]:
    d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3; tot+=d
    print(r['Kernel_Name'][23:52].ljust(30), round(d,1), r.get('Grid_Size_X') or r.get('Grid_Size'), r.get('VGPR_Count'), r.get('LDS_Block_Size'))
print('sum', tot)
PY
