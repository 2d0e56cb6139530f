# pytest (gpu) + kernel bench + bench + rocprof kernel stats of the bench; TAG = $1
set -x
mkdir -p gpurun_out
TAG=${1:-r01u}
timeout 900 python -m pytest tests -m gpu -q --no-header -rf --tb=short > gpurun_out/${TAG}_pytest.log 2>&1
echo "pytest exit $?" >> gpurun_out/${TAG}_pytest.log
tail -25 gpurun_out/${TAG}_pytest.log
timeout 300 python tools/kernel_bench.py > gpurun_out/${TAG}_kernel_bench.jsonl 2> gpurun_out/${TAG}_kernel_bench.err; grep -v xcorr gpurun_out/${TAG}_kernel_bench.jsonl; tail -3 gpurun_out/${TAG}_kernel_bench.err
timeout 400 python bench.py --no-cpu-baseline > gpurun_out/${TAG}_bench.log 2>&1; echo "bench exit $?" >> gpurun_out/${TAG}_bench.log; tail -2 gpurun_out/${TAG}_bench.log | cut -c1-400
export TMPDIR=/tmp
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_${TAG} -o ${TAG} -- python $GRAFT_REPO_ROOT/bench.py --steps 300 --warmup 30 --no-cpu-baseline --extra-streams 0 > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof_bench.log 2>&1 )
grep '"metric"' gpurun_out/${TAG}_prof_bench.log | cut -c1-300
python tools/rocpd_stats.py gpurun_out/prof_${TAG}/${TAG}_results.db --md gpurun_out/${TAG}_kernel_stats.md --title "${TAG}: bench.py --steps 300" 2>&1 | tail -3; head -14 gpurun_out/${TAG}_kernel_stats.md | cut -c1-260
