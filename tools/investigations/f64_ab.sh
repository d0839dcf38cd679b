# Pass-by-pass times of the complex128 phase correlation for one or more builds of the library (same box): VARIANTS="cur x" -> karios_amd/libkarios_hip.so,
# karios_amd/libkarios_hip_x.so; OPTS="name=value,.." development options.  Run on the GPU box: gpurun -- bash tools/investigations/f64_ab.sh
cd /tmp && export TMPDIR=/tmp
for v in ${VARIANTS:-cur}; do
  lib=$GRAFT_REPO_ROOT/karios_amd/libkarios_hip_$v.so; [ $v = cur ] && lib=$GRAFT_REPO_ROOT/karios_amd/libkarios_hip.so
  KARIOS_OPTS=$OPTS KARIOS_TIMING_ONLY=1 KARIOS_HIP_LIB=$lib timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_f64_$v -o f64 -- python3 $GRAFT_REPO_ROOT/tools/phase64_workload.py 10980 3 > /dev/null 2>&1
  echo "variant $v"; python3 $GRAFT_REPO_ROOT/tools/investigations/f64_pass_times.py $GRAFT_REPO_ROOT/gpurun_out/prof_f64_$v
done
