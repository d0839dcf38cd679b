#!/usr/bin/env python3
"""Dev tool: idle gaps between consecutive kernels of the last bench step in a rocprofv3 kernel_trace.csv."""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last step: from the last minmax_partial pair backwards
idx = [i for i, r in enumerate(rows) if "minmax_partial" in r["Kernel_Name"]]
start = idx[-4] if len(idx) >= 4 else idx[-2]
sel = rows[start:idx[-2]]   # the last but one step (the last one is followed by the tail of the run)
t0 = int(sel[0]["Start_Timestamp"])
prev_end = t0
busy = 0
for r in sel:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = s - prev_end
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][-44:]
    busy += e - s
    if gap > 3000 or e - s > 20000 or '-v' in sys.argv:
        print(f"{(s - t0) / 1e3:8.1f} us  gap {gap / 1e3:6.1f}  dur {(e - s) / 1e3:7.1f}  {name}")
    prev_end = max(prev_end, e)
print(f"span {(prev_end - t0) / 1e3:.1f} us, busy {busy / 1e3:.1f} us, kernels {len(sel)}")
