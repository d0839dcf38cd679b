#!/bin/bash
# Round-end measurements on the GPU box: the driver's bench command, the default line, the config-3 line, the rocprofv3 kernel
# statistics of the headline loop (+ full scoring), of the kernels ALONE (blocking calls), of the batched config-4 / e2e shapes, of
# config 3 and of the entry points the loop does not reach, and the PMC passes (separate --pmc runs, kernel trace only).  Results
# under gpurun_out/ (the summaries are copied to profiles/ by hand).  Every profiler run sits under `timeout`.
R=$PWD; TAG=${1:-r06}; COMMIT=${2:-unknown}
# (the LAST stdout line is the driver's record; the `detail <name> {...}` lines above it and the detail file are kept beside it)
python bench.py --gpus 1 --steps 20 --warmup 5 --detail-file gpurun_out/bench_${TAG}_detail.json > gpurun_out/bench_${TAG}.out 2> gpurun_out/bench_${TAG}.err; tail -1 gpurun_out/bench_${TAG}.out > gpurun_out/bench_${TAG}.json
python bench.py --detail-file gpurun_out/bench_${TAG}_default_detail.json > gpurun_out/bench_${TAG}_default.out 2> gpurun_out/bench_${TAG}_default.err; tail -1 gpurun_out/bench_${TAG}_default.out > gpurun_out/bench_${TAG}_default.json
python bench.py --config 3 --steps 10 --warmup 2 --detail-file gpurun_out/bench_${TAG}_config3_detail.json > gpurun_out/bench_${TAG}_config3.out 2> gpurun_out/bench_${TAG}_config3.err; tail -1 gpurun_out/bench_${TAG}_config3.out > gpurun_out/bench_${TAG}_config3.json
cd /tmp && export TMPDIR=/tmp
prof() {   # prof <tag> <script> [args]: rocprofv3 kernel statistics (csv) of `python3 <script> args`
  t=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_$t -o $t -- python3 "$@" > $R/gpurun_out/prof_${TAG}_$t.log 2>&1
}
prof c2 $R/bench.py --headline-only --steps 60 --warmup 5
prof c2drv $R/bench.py --gpus 1 --steps 20 --warmup 5 --headline-only      # (the driver's flags, the headline object alone)
prof c2one $R/bench.py --pairs-per-submission 1 --headline-only --steps 20 --warmup 5
prof c4loop $R/tools/config4_probe.py
prof alone $R/tools/blocking_workload.py 12
prof c4 $R/tools/units_probe.py config4 4
prof e2e $R/tools/units_probe.py e2e 4
prof c3 $R/bench.py --config 3 --steps 5 --warmup 1
prof ep $R/tools/entry_points_workload.py 10980 scoring,dn,auto,banded
prof f64 $R/tools/phase64_workload.py 10980 5
cd $R
sumr() { f=$(find gpurun_out/prof_${TAG}_$1 -name "$1_kernel_stats.csv" | head -1); [ -n "$f" ] && python3 tools/summarize_rocprof.py $f gpurun_out/${TAG}_kernel_stats_$2.md $3; }
sumr f64 phase_fp64 5
sumr c2 config2
sumr c2drv config2_driver_flags
sumr c2one config2_one_pair_per_submission
sumr c4loop config4_pipelined_loop
sumr alone kernels_alone 12
sumr c4 config4_batched
sumr e2e e2e_shape_batched
sumr c3 config3 6
sumr ep entry_points
# (the bench loop also runs gates and settle phases with other shapes: the PMC of the batched headline uses a stream of EXACTLY four pairs per submission)
PMC_COMMIT=$COMMIT timeout 900 bash tools/pmc_collect.sh ${TAG}_c2 tools/pairs_batched_probe.py 4
PMC_COMMIT=$COMMIT timeout 900 bash tools/pmc_collect.sh ${TAG}_ep tools/entry_points_workload.py 10980 scoring,dn
PMC_COMMIT=$COMMIT timeout 900 bash tools/pmc_collect.sh ${TAG}_c3 bench.py --config 3 --steps 4 --warmup 1
PMC_COMMIT=$COMMIT timeout 900 bash tools/pmc_collect.sh ${TAG}_f64 tools/phase64_workload.py 10980 3
# round 6: timelines of one period of the pipelined loops, the probes, the pipeline fuzz
TL_OFFSET=3 bash tools/timeline_run.sh headline > gpurun_out/${TAG}_timeline_headline.log 2>&1
TL_OFFSET=4 bash tools/timeline_run.sh c4 $R/tools/config4_probe.py > gpurun_out/${TAG}_timeline_config4.log 2>&1
python tools/two_contexts_probe.py 1 2 > gpurun_out/${TAG}_contexts_probe.log 2>&1
python tools/config4_probe.py > gpurun_out/${TAG}_config4_probe.log 2>&1
python tools/auto_ksize_probe.py > gpurun_out/${TAG}_auto_ksize.json 2> /dev/null
python tools/fuzz_pipeline.py --rounds 100000 --seed 500000 --budget-s 240 > gpurun_out/${TAG}_fuzz_pipeline.log 2>&1
tail -1 gpurun_out/bench_${TAG}.json | cut -c1-300
head -30 gpurun_out/${TAG}_kernel_stats_config2.md
head -24 gpurun_out/${TAG}_kernel_stats_kernels_alone.md
head -24 gpurun_out/${TAG}_kernel_stats_config4_batched.md
head -14 gpurun_out/${TAG}_kernel_stats_config3.md
head -16 gpurun_out/${TAG}_kernel_stats_phase_fp64.md
