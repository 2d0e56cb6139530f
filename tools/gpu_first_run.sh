set -x
mkdir -p gpurun_out
rocminfo | grep -E "Marketing Name|gfx|Compute Unit" | head -8 > gpurun_out/rocminfo.txt 2>&1
timeout 900 python -m pytest tests -m gpu -q --no-header -rA --tb=short > gpurun_out/r01_pytest.log 2>&1
echo "pytest exit $?" >> gpurun_out/r01_pytest.log
tail -60 gpurun_out/r01_pytest.log
timeout 300 python __graft_entry__.py --smoke > gpurun_out/r01_smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/r01_smoke.log; tail -5 gpurun_out/r01_smoke.log
timeout 600 python bench.py --steps 100 --warmup 10 > gpurun_out/r01_bench.log 2>&1; echo "bench exit $?" >> gpurun_out/r01_bench.log; tail -5 gpurun_out/r01_bench.log
export TMPDIR=/tmp
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r01 -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r01_prof_bench.log 2>&1 )
ls -R gpurun_out | head -40
