import csv,glob,sys
f=glob.glob(sys.argv[1]+"/**/f64_kernel_trace.csv",recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if "f64" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
tot=0
for r in rows[-10:]:
    d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1000; tot+=d
    print(r["Kernel_Name"].split("::")[-1][:30], d, r.get("Grid_Size", r.get("Grid_Size_X","")), r.get("VGPR_Count",""), r.get("LDS_Block_Size",""))
print("total",tot)
