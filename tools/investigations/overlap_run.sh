R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_ov -o ov -- python3 $R/tools/two_contexts_probe.py 2 > $R/gpurun_out/prof_ov.log 2>&1
cd $R
tail -3 gpurun_out/prof_ov.log
python tools/investigations/overlap_stats.py gpurun_out/prof_ov
rm -rf gpurun_out/prof_ov
