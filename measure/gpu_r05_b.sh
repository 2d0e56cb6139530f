# round-5 session B: GPU suite after the fixes of session A (crowd regenerated with a gentle box head, hook on the fallback
# refinement path), the early head of the tracking loop (A/B), eight ranks on one device (host contention), bench line.
#   gpurun --timeout 1500 -- 'bash measure/gpu_r05_b.sh'
TAG=r05b
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --no-header -rf --tb=short --durations=8 > gpurun_out/${TAG}_pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/${TAG}_pytest_gpu.log; tail -22 gpurun_out/${TAG}_pytest_gpu.log | cut -c1-250
grep -n "^E " gpurun_out/${TAG}_pytest_gpu.log | head -30 | cut -c1-400
timeout 300 python measure/loop_early_ab.py 30 > gpurun_out/${TAG}_loop_early_ab.jsonl 2>&1; grep '^{' gpurun_out/${TAG}_loop_early_ab.jsonl | cut -c1-220; grep -v '^{' gpurun_out/${TAG}_loop_early_ab.jsonl | tail -5
# eight ranks sharing this one device over gloo: NOT a scaling measurement — the host term of an 8-rank node (per-rank
# enqueue time under 8-way host contention, NUMA pinning)
timeout 400 python bench.py --gpus 8 --allow-shared-gpu --steps 400 --warmup 50 --no-cpu-baseline --no-parity --no-graph --extra-streams 0 --no-other-configs > gpurun_out/${TAG}_bench_8ranks_shared_gpu.log 2>&1
grep '^{' gpurun_out/${TAG}_bench_8ranks_shared_gpu.log | tail -1 > gpurun_out/${TAG}_bench_8ranks_shared_gpu.json
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r05b_bench_8ranks_shared_gpu.json"))
    print("8 ranks shared:", d["value"], d["ms_per_step"], d["per_rank"], d["config"]["parallelism"][-120:])
except Exception as e:
    print("8-rank run:", e, open("gpurun_out/r05b_bench_8ranks_shared_gpu.log").read()[-1500:])
PY
bash measure/gpu_r04_bench.sh ${TAG} | cut -c1-500
