# round-5 session A: the GPU suite with the verified order hint and the round-5 closed-loop sequences, the bench line in
# the driver's form, the cost of the hint's verification (interleaved A/B).
#   gpurun --timeout 1200 -- 'bash measure/gpu_r05_a.sh [pytest -k expression]'
TAG=r05a
mkdir -p gpurun_out
export TMPDIR=/tmp
K="$1"
if [ -n "$K" ]; then
  timeout 900 python -m pytest tests -m gpu -q --no-header -rf --tb=short -k "$K" --durations=12 > gpurun_out/${TAG}_pytest_gpu.log 2>&1
else
  timeout 900 python -m pytest tests -m gpu -q --no-header -rf --tb=short --durations=12 > gpurun_out/${TAG}_pytest_gpu.log 2>&1
fi
echo "pytest exit $?" >> gpurun_out/${TAG}_pytest_gpu.log; tail -25 gpurun_out/${TAG}_pytest_gpu.log | cut -c1-250
timeout 200 python measure/hint_ab.py 30 100 > gpurun_out/${TAG}_hint_verify_ab.jsonl 2>&1; grep '^{' gpurun_out/${TAG}_hint_verify_ab.jsonl | cut -c1-200
bash measure/gpu_r04_bench.sh ${TAG} | cut -c1-400
