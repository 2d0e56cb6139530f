set -x
mkdir -p gpurun_out
TAG=${1:-r01s}
timeout 900 python -m pytest tests -m gpu -q --no-header -rf --tb=short > gpurun_out/${TAG}_pytest.log 2>&1
echo "pytest exit $?" >> gpurun_out/${TAG}_pytest.log
tail -5 gpurun_out/${TAG}_pytest.log
timeout 400 python bench.py > gpurun_out/${TAG}_bench.log 2>&1; echo "bench exit $?" >> gpurun_out/${TAG}_bench.log; tail -2 gpurun_out/${TAG}_bench.log
timeout 400 python bench.py --tracks 100 --no-cpu-baseline > gpurun_out/${TAG}_bench_n100.log 2>&1; tail -1 gpurun_out/${TAG}_bench_n100.log
export TMPDIR=/tmp
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_${TAG} -o ${TAG} -- python $GRAFT_REPO_ROOT/bench.py --steps 500 --warmup 50 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof_bench.log 2>&1 )
bash tools/gpu_pmc.sh ${TAG} > /dev/null 2>&1
ls gpurun_out/pmc_${TAG}_*/
