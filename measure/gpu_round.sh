# One GPU-box session: pytest (gpu) + the default bench line + the driver-form bench + phase traces + rocprofv3 kernel
# stats of the bench.  TAG = $1 (outputs land in gpurun_out/${TAG}_*).  Usage on the build box:
#   gpurun --timeout 1200 -- 'bash measure/gpu_round.sh r02a'
set -x
mkdir -p gpurun_out
TAG=${1:-r02}
timeout 1200 python -m pytest tests -m gpu -q --no-header -rf --tb=short -x > gpurun_out/${TAG}_pytest.log 2>&1
echo "pytest exit $?" >> gpurun_out/${TAG}_pytest.log
tail -25 gpurun_out/${TAG}_pytest.log
timeout 400 python bench.py > gpurun_out/${TAG}_bench.log 2>&1; echo "bench exit $?" >> gpurun_out/${TAG}_bench.log; tail -2 gpurun_out/${TAG}_bench.log | cut -c1-1500
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${TAG}_bench_driver_form.log 2>&1; tail -1 gpurun_out/${TAG}_bench_driver_form.log | cut -c1-600
timeout 200 python measure/debug/fused_trace.py 30 100 > gpurun_out/${TAG}_fused_trace.jsonl 2>&1; cat gpurun_out/${TAG}_fused_trace.jsonl | cut -c1-700
export TMPDIR=/tmp
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_${TAG} -o ${TAG} -- python $GRAFT_REPO_ROOT/bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-parity --extra-streams 0 > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof_bench.log 2>&1 )
grep '"metric"' gpurun_out/${TAG}_prof_bench.log | cut -c1-300
python tools/rocpd_stats.py gpurun_out/prof_${TAG}/${TAG}_results.db --md gpurun_out/${TAG}_kernel_stats.md --title "${TAG}: bench.py --steps 300" 2>&1 | tail -3; head -16 gpurun_out/${TAG}_kernel_stats.md | cut -c1-260
