mkdir -p gpurun_out
TAG=${1:-r06d}
timeout 500 python bench.py > gpurun_out/${TAG}_bench.log 2>&1; echo "bench exit $?" >> gpurun_out/${TAG}_bench.log
python - <<PY
import json
l=[x for x in open("gpurun_out/${TAG}_bench.log") if x.startswith("{")]
d=json.loads(l[-1])
print("value", d["value"], "ms/step", d["ms_per_step"], "host", d["per_rank"]["host_enqueue_us_per_step_max"])
print("roofline", {k: d["roofline"].get(k) for k in ("avg_launch_us","frac","achieved")})
print("tower", d.get("roofline_tower"))
for k, v in (d.get("other_configs") or {}).items():
    print(k, {kk: v.get(kk) for kk in ("ms_per_step","value")}, (v.get("roofline") or {}).get("avg_launch_us"), (v.get("roofline") or {}).get("frac"), (v.get("roofline_tower") or {}).get("avg_launch_us"), v.get("parity"))
tl = d.get("tracking_loop") or {}
print("loop", tl.get("ms_per_frame"), (tl.get("with_refinement") or {}).get("ms_per_frame"), (tl.get("next_frame_shown") or {}).get("ms_per_frame"))
print("cpu", d.get("cpu_baseline"))
print("parity", json.dumps(d.get("parity"))[:600])
PY
timeout 300 python tools/aot_bench.py > gpurun_out/${TAG}_aot.log 2>&1; tail -3 gpurun_out/${TAG}_aot.log | cut -c1-600
