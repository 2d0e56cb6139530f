# round-5 session D: the closed-loop device replays with arg-max ties aligned to the reference's side (tests/sequence_replay.py).
#   gpurun --timeout 900 -- 'bash measure/gpu_r05_d.sh'
TAG=r05d
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 800 python -m pytest tests/test_sequence.py tests/test_solver.py -m gpu -q --no-header -rf --tb=short -s > gpurun_out/${TAG}_pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/${TAG}_pytest_gpu.log; tail -8 gpurun_out/${TAG}_pytest_gpu.log | cut -c1-250
grep -n "^E " gpurun_out/${TAG}_pytest_gpu.log | head -20 | cut -c1-700
grep -n "closed loop crowd\|closed loop longdormant\|closed loop multiclass\|fallbacks taken" gpurun_out/${TAG}_pytest_gpu.log | cut -c1-900 | tail -40
