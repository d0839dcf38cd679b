R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_blk -o p -- python3 $R/tools/blocking_workload.py 6 > /dev/null 2>&1
echo rc=$?
cd $R
python3 - <<'PY'
import csv, collections, glob
f = glob.glob('gpurun_out/pmc_blk/**/p_counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].split('(')[0][-40:]
    acc[k][r['Counter_Name']] += float(r['Counter_Value']); n[(k, r['Counter_Name'])] += 1
for k in acc:
    if any(s in k for s in ('lk2', 'eig3', 'lap_march', 'zncc_int', 'f_hist', 'f_sweep')):
        print(k, {c: round(v / n[(k, c)]) for c, v in acc[k].items()})
PY
