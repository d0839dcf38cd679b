R=$PWD
mkdir -p $R/gpurun_out/prof_c3; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c3 -o p -- python3 $R/tools/config3_time.py > $R/gpurun_out/prof_c3/run.log 2>&1
cd $R
python3 - <<'PY'
import csv, collections
acc=collections.defaultdict(list)
for r in csv.DictReader(open("gpurun_out/prof_c3/p_kernel_trace.csv")):
    acc[r["Kernel_Name"][:70]].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,v in sorted(acc.items(), key=lambda kv:-sum(kv[1]))[:14]:
    print(f"{k:70s} n={len(v):3d} avg={sum(v)/len(v):9.1f} us min={min(v):9.1f}")
PY
