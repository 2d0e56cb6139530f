# the driver's form of the bench (N=1, --steps 20) + a 2000-step run
mkdir -p gpurun_out
TAG=${1:-r04}
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_driver_form.log 2>&1
python - <<PY
import json
l=[x for x in open("gpurun_out/${TAG}_bench_driver_form.log") if x.startswith("{")]
if l:
    d=json.loads(l[-1]); open("gpurun_out/${TAG}_bench_line_driver_form.json","w").write(l[-1])
    print("value %.0f ms/step %.5f fused_us %.2f frac %.4f tower %s" % (d["value"], d["ms_per_step"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"], (d.get("roofline_tower") or {}).get("avg_launch_us")))
    print("traffic", d["roofline"]["traffic"], d["roofline"]["traffic_source"][:120])
    print("fallbacks", d.get("fallbacks"))
    for k,v in (d.get("other_configs") or {}).items():
        print(k, {kk: v.get(kk) for kk in ("value","ms_per_step","error")}, "fused frac", (v.get("roofline") or {}).get("frac"), "us", (v.get("roofline") or {}).get("avg_launch_us"), "tower", (v.get("roofline_tower") or {}).get("avg_launch_us"), "golden", ((v.get("parity") or {}).get("vs_reference_golden") or {}).get("argmax_exact_frac"), "fallbacks", v.get("fallbacks"))
    print("loop", (d.get("tracking_loop") or {}).get("ms_per_frame"), ((d.get("tracking_loop") or {}).get("with_refinement") or {}).get("ms_per_frame"))
    print("cpu", d.get("cpu_baseline"))
else:
    print(open("gpurun_out/${TAG}_bench_driver_form.log").read()[-2500:])
PY
