# full GPU suite (no -x: see everything that broke) + the driver-form bench line + rocprofv3 kernel stats of the bench
mkdir -p gpurun_out
TAG=${1:-r06full}
timeout 1500 python -m pytest tests -m gpu -q --no-header -rf --tb=short > gpurun_out/${TAG}_pytest.log 2>&1
echo "pytest exit $?" >> gpurun_out/${TAG}_pytest.log
grep -n "passed\|failed\|^FAILED\|^ERROR" gpurun_out/${TAG}_pytest.log | cut -c1-260 | tail -40
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${TAG}_bench_driver_form.log 2>&1; tail -1 gpurun_out/${TAG}_bench_driver_form.log | cut -c1-900
export TMPDIR=/tmp
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_${TAG} -o ${TAG} -- python $GRAFT_REPO_ROOT/bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-parity --extra-streams 0 > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof_bench.log 2>&1 )
grep '"metric"' gpurun_out/${TAG}_prof_bench.log | cut -c1-300
python tools/rocpd_stats.py gpurun_out/prof_${TAG}/${TAG}_results.db --md gpurun_out/${TAG}_kernel_stats.md --title "${TAG}: bench.py --steps 300" 2>&1 | tail -3; head -16 gpurun_out/${TAG}_kernel_stats.md | cut -c1-260
rm -rf gpurun_out/prof_${TAG}
