# round-5 session C: GPU suite after the host-chain work (lean TrackingLoop.__call__, steady-frame fast path of
# _finish_frame, output views before the record) and the relaxed IoU floor of flipped crowd tracks; the loop A/B; bench line.
#   gpurun --timeout 1200 -- 'bash measure/gpu_r05_c.sh'
TAG=r05c
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --no-header -rf --tb=short --durations=6 > gpurun_out/${TAG}_pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/${TAG}_pytest_gpu.log; tail -16 gpurun_out/${TAG}_pytest_gpu.log | cut -c1-250
grep -n "^E " gpurun_out/${TAG}_pytest_gpu.log | head -30 | cut -c1-600
timeout 300 python measure/loop_early_ab.py 30 > gpurun_out/${TAG}_loop_early_ab.jsonl 2>&1; grep '^{' gpurun_out/${TAG}_loop_early_ab.jsonl | cut -c1-220; grep -v '^{' gpurun_out/${TAG}_loop_early_ab.jsonl | tail -5
bash measure/gpu_r04_bench.sh ${TAG} | cut -c1-500
