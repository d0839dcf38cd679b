import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "minmax_partial" in r["Kernel_Name"]]
start = idx[-4]; end = idx[-2]
sel = rows[start:end]
t0 = int(sel[0]["Start_Timestamp"]); prev=t0
for r in sel:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][-50:]
    print(f"{(s - t0) / 1e3:8.1f} us  gap {(s-prev) / 1e3:6.1f}  dur {(e - s) / 1e3:7.1f}  {name}")
    prev=max(prev,e)
