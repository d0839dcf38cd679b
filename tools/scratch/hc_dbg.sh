cd /tmp && export TMPDIR=/tmp
for v in 0 1 2 4 7; do
  export HC_DBG=$v
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/hc_$v -o h -- python3 $GRAFT_REPO_ROOT/tools/scratch/stall_hunt.py 60 1 > /dev/null 2>&1
  f=$(find $GRAFT_REPO_ROOT/gpurun_out/hc_$v -name 'h_kernel_stats.csv' | head -1)
  echo "HC_DBG=$v $(grep -E 'f_hist_cut|f_scatter_cells' $f | awk -F, '{print $1, $4}' | tr '\n' ' ')"
done
