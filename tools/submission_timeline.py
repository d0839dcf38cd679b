#!/usr/bin/env python3
"""Timeline of ONE batched submission (kernel start, gap to the previous end on any queue, duration, queue) from a rocprofv3 kernel trace:
python tools/submission_timeline.py <rocprofv3 output dir>."""
import csv,sys,glob
f=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True)[0]
rows=[r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(rows) if 'lap_march_units_kernel' in r['Kernel_Name']]
k=len(idx)//2+(int(sys.argv[2]) if len(sys.argv)>2 else 0)
a,b=idx[k],idx[k+1]
t0=int(rows[a]['Start_Timestamp']); prev=t0
print('submission span us', (int(rows[b]['Start_Timestamp'])-t0)/1e3)
for r in rows[a:b]:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    name=r['Kernel_Name'].replace('(anonymous namespace)::','').split('(')[0][-36:]
    print(f"{(s-t0)/1e3:8.1f} gap {(s-prev)/1e3:7.1f} dur {(e-s)/1e3:7.1f} q{r.get('Queue_Id','')} {name}")
    prev=max(prev,e)
