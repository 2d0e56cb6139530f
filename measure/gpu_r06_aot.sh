mkdir -p gpurun_out
TAG=${1:-r06aot}
export TMPDIR=/tmp
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_${TAG} -o ${TAG} -- python $GRAFT_REPO_ROOT/tools/aot_bench.py > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof.log 2>&1 )
tail -1 gpurun_out/${TAG}_prof.log | cut -c1-300
python tools/rocpd_stats.py gpurun_out/prof_${TAG}/${TAG}_results.db --md gpurun_out/${TAG}_kernel_stats.md --title "${TAG}: tools/aot_bench.py" 2>&1 | tail -2; head -14 gpurun_out/${TAG}_kernel_stats.md | cut -c1-230
rm -rf gpurun_out/prof_${TAG}
