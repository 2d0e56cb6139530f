# round 4, last session: the whole GPU suite and the bench lines at HEAD
TAG=r04
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 400 python -m pytest tests -m gpu -q --no-header -rf --tb=short > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/${TAG}_pytest_gpu.log; tail -3 gpurun_out/${TAG}_pytest_gpu.log | cut -c1-200
timeout 300 python bench.py > gpurun_out/${TAG}_bench.log 2>&1; echo "bench exit $?"; tail -1 gpurun_out/${TAG}_bench.log > gpurun_out/${TAG}_bench_line.json; cut -c1-260 gpurun_out/${TAG}_bench_line.json
bash measure/gpu_r04_bench.sh ${TAG} | cut -c1-260
