set -x
python -m pytest tests/test_hip_parity.py -m gpu -q --no-header --tb=short -x -k "decode or emm or benchmark" 2>&1 | tail -6
python tools/debug/decode_trace.py 30 100
python bench.py --no-cpu-baseline --extra-streams 0 > gpurun_out/r02t_bench.log 2>&1; python - <<'PY'
import json
for l in open('gpurun_out/r02t_bench.log'):
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline_tower']['avg_launch_us'], d['parity']['vs_reference_golden'])
PY
