#!/usr/bin/env python3
"""Which submission does a kernel belong to?  The j-th launch of a stage's kernel in a window IS submission j: lists, per submission of the
last window of a rocprofv3 kernel trace, start / end of its min-max, Laplacian pass, LK and eigenvalue pass - who waits for whom.
python tools/investigations/stage_order.py <rocprofv3 output dir>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
wins, cur, end = [], [], None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if end is not None and s - end > 1_000_000:
        wins.append(cur); cur = []
    cur.append(r); end = e if end is None else max(end, e)
wins.append(cur)
wins = [w for w in wins if sum('lap_march_units_kernel' in r['Kernel_Name'] for r in w) >= 4]
w = wins[-1]
t0 = int(w[0]["Start_Timestamp"])
names = {"mm": "minmax_partial_units", "L": "lap_march_units", "K": "lk2_units", "E": "eig3_units", "P": "pyrdown_units", "Z": "zncc_int_units"}
seq = {k: [(int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0) for r in w if v in r["Kernel_Name"]] for k, v in names.items()}
n = len(seq["L"])
print("submission:  " + "  ".join(f"{k:>15s}" for k in names))
for j in range(n):
    print(f"{j:10d}:  " + "  ".join((f"{seq[k][j][0] / 1e3:7.0f}-{seq[k][j][1] / 1e3:7.0f}" if j < len(seq[k]) else " " * 15) for k in names))
