export TMPDIR=/tmp
for n in 30 100; do for v in 1 2 4; do
  export SMOT_DECODE_SPLIT=$v
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_ds${n}_$v -o ds -- python $GRAFT_REPO_ROOT/bench.py --tracks $n --steps 300 --warmup 30 --no-cpu-baseline --extra-streams 0 > /dev/null 2>&1)
  echo "N=$n split=$v: $(python tools/rocpd_stats.py gpurun_out/prof_ds${n}_$v/ds_results.db | grep decode_band | awk -F'|' '{print $4}')"
done; done
