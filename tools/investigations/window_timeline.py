#!/usr/bin/env python3
"""Timeline of ONE whole window of the pipelined loop (from an idle GPU to an idle GPU: fill, steady periods, drain) from a rocprofv3
kernel trace: the dense kernels and every idle gap of the main queue - where the fixed cost of a timed region goes.
python tools/investigations/window_timeline.py <rocprofv3 output dir> [window index from the end, default 1]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# windows: separated by gaps > 1.5 ms with no kernel running on any queue
wins, cur, end = [], [], None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if end is not None and s - end > 1_500_000:
        wins.append(cur); cur = []
    cur.append(r); end = e if end is None else max(end, e)
wins.append(cur)
wins = [w for w in wins if sum('lap_march_units_kernel' in r['Kernel_Name'] for r in w) >= 3]
w = wins[-int(sys.argv[2]) if len(sys.argv) > 2 else -1]
t0 = int(w[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in w)
nl = sum('lap_march_units_kernel' in r['Kernel_Name'] for r in w)
print(f"window: {len(w)} kernels, {nl} submissions, {(t1 - t0) / 1e3:.1f} us from the first kernel's start to the last one's end")
busy_end = t0
big = ('lap_march_units', 'eig3_units', 'lk2_units', 'partial_units', 'pyrdown_units', 'ncc_int_units', 'f_sweep')
for r in w:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0][-34:]
    idle = s - busy_end
    if idle > 20_000:
        print(f"{(busy_end - t0) / 1e3:9.1f}   -- no kernel on any queue for {idle / 1e3:.1f} us")
    if any(b in name for b in big) and (e - s) > 150_000 or idle > 20_000:
        print(f"{(s - t0) / 1e3:9.1f} dur {(e - s) / 1e3:7.1f} q{r.get('Queue_Id', '')} {name}")
    busy_end = max(busy_end, e)
