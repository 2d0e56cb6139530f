# round-5 session E: the order hint also ranks the dormant rows carried inside the solver's launch (memories with dormant
# tracks keep their hint) — sequence / solver / hint tests, and the dormant-track loop A/B.
#   gpurun --timeout 900 -- 'bash measure/gpu_r05_e.sh'
TAG=r05e
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 800 python -m pytest tests/test_sequence.py tests/test_solver.py tests/test_hip_parity.py -m gpu -q --no-header -rf --tb=short -s -k "sequence or solver or hint or dormant or closed_loop or loop" > gpurun_out/${TAG}_pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/${TAG}_pytest_gpu.log; tail -6 gpurun_out/${TAG}_pytest_gpu.log | cut -c1-250
grep -n "^E " gpurun_out/${TAG}_pytest_gpu.log | head -20 | cut -c1-700
grep -n "fallbacks taken" gpurun_out/${TAG}_pytest_gpu.log | cut -c1-300 | tail -34
timeout 300 python measure/loop_early_ab.py 30 > gpurun_out/${TAG}_loop_early_ab.jsonl 2>&1; grep '^{' gpurun_out/${TAG}_loop_early_ab.jsonl | cut -c1-220; grep -v '^{' gpurun_out/${TAG}_loop_early_ab.jsonl | tail -5
