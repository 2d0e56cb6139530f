export TMPDIR=/tmp
for v in 0 1 2; do
  export SMOT_FUSED_ABL=$v
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_fa$v -o fa -- python $GRAFT_REPO_ROOT/bench.py --steps 300 --warmup 30 --no-cpu-baseline --extra-streams 0 > /dev/null 2>&1)
  echo "fused abl=$v: $(python tools/rocpd_stats.py gpurun_out/prof_fa$v/fa_results.db | grep 'fused8_kernel<30' | awk -F'|' '{print $4, $5}')"
done
