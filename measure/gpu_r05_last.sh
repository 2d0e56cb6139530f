# round-5 last session: GPU suite + both bench lines at HEAD.
TAG=r05
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --no-header -rf --tb=short > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/${TAG}_pytest_gpu.log; tail -4 gpurun_out/${TAG}_pytest_gpu.log | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 700 python bench.py > gpurun_out/${TAG}_bench.log 2>&1; echo "bench exit $?"; tail -1 gpurun_out/${TAG}_bench.log > gpurun_out/${TAG}_bench_line.json; cut -c1-200 gpurun_out/${TAG}_bench_line.json
( time bash measure/gpu_r04_bench.sh ${TAG} ) 2>&1 | cut -c1-400
