# round 4: dormant rows carried on the device (smot_memory_carry_fwd) — tests, path-equivalence soaks, loop timings
#   gpurun --timeout 900 -- 'bash measure/gpu_r04_carry.sh'
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q --no-header -rf --tb=short -x > gpurun_out/r04_pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/r04_pytest_gpu.log; tail -15 gpurun_out/r04_pytest_gpu.log | cut -c1-300
( timeout 120 python measure/debug/loop_equiv_soak.py 200; echo "exit $?"
  SWITCH=1 timeout 120 python measure/debug/loop_equiv_soak.py 200; echo "exit $?"
  AHEAD=1 PEEK=5 MAXD=30 timeout 120 python measure/debug/loop_equiv_soak.py 200; echo "exit $?"
  AHEAD=1 PEEK=7 MAXD=8 SEED=3 timeout 120 python measure/debug/loop_equiv_soak.py 200; echo "exit $?" ) > gpurun_out/r04_carry_soak.log 2>&1
grep -v "^$" gpurun_out/r04_carry_soak.log | tail -12 | cut -c1-400
timeout 400 python bench.py --no-cpu-baseline --no-parity --no-graph --no-other-configs > gpurun_out/r04_carry_bench.log 2>&1; echo "bench exit $?"
tail -1 gpurun_out/r04_carry_bench.log > gpurun_out/r04_carry_bench_line.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04_carry_bench_line.json"))
t = d["tracking_loop"]
print("loop", t["ms_per_frame"], "shown", t["next_frame_shown"]["ms_per_frame"])
w = t["with_dormant_tracks"]
print("dormant: device", w["ms_per_frame"], {k: w[k] for k in ("active_tracks", "dormant_tracks", "memory_rows", "track_count_held")})
print("dormant: host form", w["host_form"]["ms_per_frame"], "shown", w["next_frame_shown"]["ms_per_frame"], w["next_frame_shown"]["speculative_heads"])
print("value", d["value"], "ms", d["ms_per_step"])
PY
