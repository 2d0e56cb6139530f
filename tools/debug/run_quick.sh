set -x
python -m pytest tests -m gpu -q --no-header --tb=short -x -k "timer or bench or host" 2>&1 | tail -3
python bench.py --no-cpu-baseline --extra-streams 0 2>&1 | tail -1 > gpurun_out/q_bench.json; python - <<'PY'
import json
d=json.loads(open('gpurun_out/q_bench.json').read())
print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['frac'], d['roofline']['launches_timed'], d['roofline']['xcorr_op']['avg_launch_us'], d['roofline']['xcorr_op']['frac'], d['roofline_tower']['avg_launch_us'], d['roofline_tower']['frac'])
PY
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-300
ABLS=0 OCTS=2 python tools/debug/tower_bench.py 2>&1 | grep "^{" | head -2 | cut -c1-200
