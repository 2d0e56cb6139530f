# round-5 session G: the bounded loop soak and the GPU suite at HEAD.
TAG=r05
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python measure/debug/loop_soak_r05.py 4 > gpurun_out/${TAG}_loop_soak.log 2>&1; echo "soak exit $?" >> gpurun_out/${TAG}_loop_soak.log; tail -9 gpurun_out/${TAG}_loop_soak.log | cut -c1-700
timeout 900 python -m pytest tests -m gpu -q --no-header -rf --tb=short > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/${TAG}_pytest_gpu.log; tail -4 gpurun_out/${TAG}_pytest_gpu.log | cut -c1-200
