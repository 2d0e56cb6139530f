# round 4: where tower_gn_heads_kernel<29> spends its 20 us (phase trace), + a bench line after the fallbacks attribution edit
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 120 python measure/debug/gn_heads_trace.py 30 > gpurun_out/r04_gn_heads_trace.jsonl 2>&1; grep '^{' gpurun_out/r04_gn_heads_trace.jsonl | cut -c1-600 || tail -5 gpurun_out/r04_gn_heads_trace.jsonl
timeout 200 python -m pytest tests/test_hip_parity.py -m gpu -q --no-header -x -k "blocked or aot or second_yaml or 29" 2>&1 | tail -3
timeout 200 python bench.py --steps 300 --no-cpu-baseline --no-parity --no-graph --no-other-configs > gpurun_out/r04_bench_quick.log 2>&1; echo "bench exit $?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04_bench_quick.log").read().strip().splitlines()[-1])
print("fallbacks", d.get("fallbacks"), "dormant fallbacks", d["tracking_loop"]["with_dormant_tracks"].get("fallbacks"))
print("value", d["value"], "loop", d["tracking_loop"]["ms_per_frame"], d["tracking_loop"]["with_dormant_tracks"]["ms_per_frame"], d["tracking_loop"]["with_dormant_tracks"]["next_frame_shown"]["ms_per_frame"])
PY
