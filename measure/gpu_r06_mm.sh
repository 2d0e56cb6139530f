mkdir -p gpurun_out
timeout 300 python measure/debug/fused_mm_check.py > gpurun_out/r06_fused_mm_check.jsonl 2>&1
cat gpurun_out/r06_fused_mm_check.jsonl | tail -8
for N in 30 100; do
timeout 300 python bench.py --tracks $N --steps 300 --warmup 30 --no-cpu-baseline > gpurun_out/r06mm_bench_n$N.log 2>&1
python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r06mm_bench_n$N.log") if l.startswith("{")][-1])
print($N, d["value"], d["ms_per_step"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"], (d.get("roofline_tower") or {}).get("avg_launch_us"), d["parity"])
PY
done
