# rocprofv3 kernel trace of a workload + the timeline of one period (tools/submission_timeline.py) + kernel statistics.
# usage: bash tools/timeline_run.sh <tag> <script> [args...]     (default: the headline loop of bench.py)
R=$PWD; TAG=${1:-a}; shift
if [ $# -eq 0 ]; then set -- $R/bench.py --headline-only --steps 20 --warmup 5; fi
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_tl -o tl -- python3 "$@" > $R/gpurun_out/prof_tl.log 2>&1
cd $R
python tools/submission_timeline.py gpurun_out/prof_tl ${TL_OFFSET:-0} > gpurun_out/timeline_r06_$TAG.txt
f=$(find gpurun_out/prof_tl -name "tl_kernel_stats.csv" | head -1); python3 tools/summarize_rocprof.py $f gpurun_out/r06${TAG}_kernel_stats.md
cat gpurun_out/timeline_r06_$TAG.txt; head -24 gpurun_out/r06${TAG}_kernel_stats.md
rm -rf gpurun_out/prof_tl
