#!/bin/bash
# PMC passes (separate rocprofv3 --pmc runs, kernel trace only) over BLOCKING tile calls: every kernel alone.  Prints per-launch averages of
# the kernels named in $KERNELS (default: the big ones).  usage: bash tools/pmc_blocking.sh "<counters pass 1>" ["<counters pass 2>" ...]
R=$PWD
KERNELS=${KERNELS:-lk2 eig3 lap_march zncc_int f_hist f_sweep}
cd /tmp && export TMPDIR=/tmp
i=0
for pass in "$@"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $R/gpurun_out/pmc_blk$i -o p -- python3 $R/tools/blocking_workload.py 6 > /dev/null 2>&1
  echo "pass $i ($pass) rc=$?"
  python3 - "$R/gpurun_out/pmc_blk$i" $KERNELS <<'PY'
import csv, collections, glob, sys
f = glob.glob(sys.argv[1] + '/**/p_counter_collection.csv', recursive=True)[0]
want = sys.argv[2:]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].split('(')[0][-44:]
    acc[k][r['Counter_Name']] += float(r['Counter_Value']); n[(k, r['Counter_Name'])] += 1
for k in acc:
    if any(s in k for s in want):
        print(' ', k, {c: round(v / n[(k, c)]) for c, v in acc[k].items()})
PY
done
