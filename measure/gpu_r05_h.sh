# round-5 session H: GPU suite at HEAD after the early head's lifetime fix (twice: the failure it fixes depended on what the
# session's other tests had left in ops' caches).
TAG=r05
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --no-header -rf --tb=short > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/${TAG}_pytest_gpu.log; tail -4 gpurun_out/${TAG}_pytest_gpu.log | cut -c1-200
timeout 900 python -m pytest tests -m gpu -q --no-header -rf --tb=short -p no:randomly > gpurun_out/${TAG}_pytest_gpu_2.log 2>&1; echo "pytest exit $?" >> gpurun_out/${TAG}_pytest_gpu_2.log; tail -4 gpurun_out/${TAG}_pytest_gpu_2.log | cut -c1-200
grep -n "^E " gpurun_out/${TAG}_pytest_gpu.log gpurun_out/${TAG}_pytest_gpu_2.log | head -10 | cut -c1-400
