# session for the order hint: its tests, the closed loop, the A/B, bench lines
TAG=${1:-hint}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_sequence.py tests/test_solver.py -m gpu -q --no-header -rf --tb=short -x > gpurun_out/${TAG}_pytest.log 2>&1
echo "pytest exit $?" >> gpurun_out/${TAG}_pytest.log
grep -n "passed\|failed\|^E  " gpurun_out/${TAG}_pytest.log | cut -c1-600 | tail -12
timeout 400 python measure/hint_ab.py 30 100 > gpurun_out/${TAG}_ab.jsonl 2> gpurun_out/${TAG}_ab.err; cat gpurun_out/${TAG}_ab.jsonl; tail -3 gpurun_out/${TAG}_ab.err
for n in 30; do
  timeout 300 python bench.py --no-cpu-baseline --tracks $n --steps 1000 > gpurun_out/${TAG}_bench_n$n.log 2>&1
  grep '^{' gpurun_out/${TAG}_bench_n$n.log | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('value %.0f ms/step %.5f fused_us %.2f frac %.4f tower_us %s loop %s' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['frac'], (d.get('roofline_tower') or {}).get('avg_launch_us'), json.dumps(d.get('tracking_loop'))[:300]))"
done
