cd /tmp; export TMPDIR=/tmp
for v in 12 24 48 96 160; do
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_crowd
  PROBE_CASES=crowd PROBE_CROWD_VALUES=$v rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_crowd -o r -- python $GRAFT_REPO_ROOT/tools/gpu_depth_distribution_probe.py > /tmp/out_$v.log 2>&1
  echo "values $v: $(grep '^{' /tmp/out_$v.log | cut -c1-200)"
  grep -E "ds_segsort|ds_hist|ds_scatter|rs_scatter" $GRAFT_REPO_ROOT/gpurun_out/prof_crowd/r_kernel_stats.csv | sed -E 's/\(unsigned.*\)",/",/' | cut -c1-160
done
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_crowd
