#!/usr/bin/env python3
"""From a rocprofv3 kernel trace of several contexts in flight: duration of every latency-bound chain kernel when a DENSE kernel of another
queue was running for its whole span, against when none was.  python tools/investigations/overlap_stats.py <rocprofv3 output dir>"""
import csv, glob, sys, collections, bisect
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
DENSE = ('lap_march', 'eig3_', 'lk2_')
def short(n): return n.replace('(anonymous namespace)::', '').split('(')[0].split('<')[0][-28:]
ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Queue_Id'], short(r['Kernel_Name'])) for r in rows]
dense = sorted((s, e, q, n) for s, e, q, n in ev if any(d in n for d in DENSE))
stats = collections.defaultdict(lambda: {'alone': [], 'beside': []})
for s, e, q, n in ev:
    if any(d in n for d in DENSE) or n.startswith('at::') or 'rocclr' in n:
        continue
    cover = 0
    for ds, de, dq, dn in dense:
        if dq != q and ds < e and de > s:
            cover += min(e, de) - max(s, ds)
    frac = cover / max(1, e - s)
    if frac > 0.9: stats[n]['beside'].append((e - s) / 1e3)
    elif frac < 0.1: stats[n]['alone'].append((e - s) / 1e3)
print(f"{'kernel':30s} {'alone n':>8s} {'med us':>8s} {'beside n':>9s} {'med us':>8s}")
med = lambda v: sorted(v)[len(v) // 2] if v else float('nan')
for n, d in sorted(stats.items()):
    print(f"{n:30s} {len(d['alone']):8d} {med(d['alone']):8.1f} {len(d['beside']):9d} {med(d['beside']):8.1f}")
