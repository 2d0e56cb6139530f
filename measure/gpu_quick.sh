# quick GPU session: the parity tests that touch the pooling / correlation / head kernels + bench lines at 30 and 100 tracks
# usage: gpurun --timeout 900 -- 'bash measure/gpu_quick.sh TAG'
TAG=${1:-q}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_hip_parity.py tests/test_sequence.py -m gpu -q --no-header -rf --tb=short -x -k "${KEXPR:-fused or pool or benchmark or emm or closed or roi}" > gpurun_out/${TAG}_pytest.log 2>&1
echo "pytest exit $?" >> gpurun_out/${TAG}_pytest.log
grep -n "passed\|failed\|^E  " gpurun_out/${TAG}_pytest.log | cut -c1-600 | tail -12
for n in ${TRACKS:-30 100}; do
  timeout 300 python bench.py --no-cpu-baseline --tracks $n --steps 1000 > gpurun_out/${TAG}_bench_n$n.log 2>&1
  python - <<PY
import json
l=[x for x in open("gpurun_out/${TAG}_bench_n$n.log") if x.startswith("{")]
if l:
    d=json.loads(l[-1]); pr=d.get("parity") or {}
    print("N=$n value %.0f ms/step %.5f fused_us %.2f frac %.4f tower_us %s argmax %s" % (d["value"], d["ms_per_step"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"], (d.get("roofline_tower") or {}).get("avg_launch_us"), (pr.get("vs_oracle_fp32") or {}).get("argmax_exact_frac")))
else:
    print(open("gpurun_out/${TAG}_bench_n$n.log").read()[-1500:])
PY
done
