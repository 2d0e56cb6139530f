set -x
mkdir -p gpurun_out
TAG=${1:-r01d}
timeout 900 python -m pytest tests -m gpu -q --no-header -rf --tb=short > gpurun_out/${TAG}_pytest.log 2>&1
echo "pytest exit $?" >> gpurun_out/${TAG}_pytest.log
tail -15 gpurun_out/${TAG}_pytest.log
export TMPDIR=/tmp
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}_kb -o kb -- python $GRAFT_REPO_ROOT/tools/kernel_bench.py > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_kernel_bench.jsonl 2> $GRAFT_REPO_ROOT/gpurun_out/${TAG}_kernel_bench.err )
cat gpurun_out/${TAG}_kernel_bench.jsonl | cut -c1-220
timeout 400 python bench.py --no-cpu-baseline > gpurun_out/${TAG}_bench.log 2>&1; echo "bench exit $?" >> gpurun_out/${TAG}_bench.log; tail -2 gpurun_out/${TAG}_bench.log | cut -c1-330
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_${TAG} -o ${TAG} -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof_bench.log 2>&1 )
grep '"metric"' gpurun_out/${TAG}_prof_bench.log | cut -c1-300
rocprofv3 -L > gpurun_out/counters_gfx950.txt 2>&1
wc -l gpurun_out/counters_gfx950.txt
