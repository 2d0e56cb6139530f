set -x
mkdir -p gpurun_out
TAG=${1:-tw}
timeout 300 python -m pytest tests -m gpu -q --no-header -x -k "predictor or tower or configs" 2>&1 | tail -4
timeout 200 python tools/debug/tower_bench.py 2>&1 | grep tracks
export TMPDIR=/tmp
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_${TAG} -o ${TAG} -- python $GRAFT_REPO_ROOT/tools/debug/tower_bench.py > /dev/null 2>&1 )
python tools/rocpd_stats.py gpurun_out/prof_${TAG}/${TAG}_results.db --by-grid 2>&1 | grep -i "tower\|combine" | cut -c1-250
