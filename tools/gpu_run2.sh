set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --no-header -rA --tb=short > gpurun_out/r01b_pytest.log 2>&1
echo "pytest exit $?" >> gpurun_out/r01b_pytest.log
tail -45 gpurun_out/r01b_pytest.log
timeout 300 python tools/kernel_bench.py > gpurun_out/r01b_kernel_bench.jsonl 2> gpurun_out/r01b_kernel_bench.err; cat gpurun_out/r01b_kernel_bench.jsonl; tail -3 gpurun_out/r01b_kernel_bench.err
timeout 400 python bench.py > gpurun_out/r01b_bench.log 2>&1; echo "bench exit $?" >> gpurun_out/r01b_bench.log; tail -3 gpurun_out/r01b_bench.log
export TMPDIR=/tmp
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r01b -o r01b -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r01b_prof_bench.log 2>&1 )
grep '"metric"' gpurun_out/r01b_prof_bench.log | head -c 1500
