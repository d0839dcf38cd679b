R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_tl -o tl -- python3 $R/bench.py --headline-only --steps 20 --warmup 5 > $R/gpurun_out/prof_tl.log 2>&1
cd $R
python tools/submission_timeline.py gpurun_out/prof_tl > gpurun_out/timeline_r06_a.txt
f=$(find gpurun_out/prof_tl -name "tl_kernel_stats.csv" | head -1); python3 tools/summarize_rocprof.py $f gpurun_out/r06a_kernel_stats.md
cat gpurun_out/timeline_r06_a.txt; head -40 gpurun_out/r06a_kernel_stats.md
rm -rf gpurun_out/prof_tl
