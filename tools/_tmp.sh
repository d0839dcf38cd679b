R=$PWD; cd /tmp && export TMPDIR=/tmp
KARIOS_PROBE_SUBS=8 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_so -o w -- python3 $R/tools/pairs_batched_probe.py 4 > $R/gpurun_out/prof_so.log 2>&1
cd $R; python tools/stage_order.py gpurun_out/prof_so; python tools/window_timeline.py gpurun_out/prof_so | head -3; rm -rf gpurun_out/prof_so
