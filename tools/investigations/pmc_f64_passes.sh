#!/bin/bash
# PMC passes over the complex128 phase correlation, printed PER DISPATCH in launch order for the last call (the nine level passes of a
# call share two kernel names).  usage: bash tools/investigations/pmc_f64_passes.sh "<counters pass 1>" ["<counters pass 2>" ...]
R=$PWD
cd /tmp && export TMPDIR=/tmp
i=0
for pass in "$@"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $R/gpurun_out/pmc_f64p$i -o p -- python3 $R/tools/phase64_workload.py 10980 2 > /dev/null 2>&1
  echo "pass $i ($pass) rc=$?"
  python3 - "$R/gpurun_out/pmc_f64p$i" <<'PY'
import csv, collections, glob, sys
f = glob.glob(sys.argv[1] + '/**/p_counter_collection.csv', recursive=True)[0]
d = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if 'f64' not in r['Kernel_Name']: continue
    k = int(r['Dispatch_Id'])
    d.setdefault(k, [r['Kernel_Name'].split('::')[-1][:24], {}])[1][r['Counter_Name']] = round(float(r['Counter_Value']))
ids = sorted(d)[-10:]
for k in ids: print(' ', k, d[k][0], d[k][1])
PY
done
