# round-5 session I: the dormant256 sequence (> 256 dormant rows: host concatenation fallback) through the four device paths.
TAG=r05i
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 800 python -m pytest tests/test_sequence.py -m gpu -q --no-header -rf --tb=short -s -k "dormant256" > gpurun_out/${TAG}_pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/${TAG}_pytest_gpu.log; tail -5 gpurun_out/${TAG}_pytest_gpu.log | cut -c1-250
grep -n "^E " gpurun_out/${TAG}_pytest_gpu.log | head -20 | cut -c1-900
grep -n "closed loop dormant256\|fallbacks taken" gpurun_out/${TAG}_pytest_gpu.log | cut -c1-1500 | tail -10
