# scratch: one GPU-box session for the kernel under work (edit freely; tools/gpu_round.sh is the full round)
set -x
python -m pytest tests/test_hip_parity.py -m gpu -q --no-header --tb=short -x -k "predictor or tower or emm" 2>&1 | tail -3
for n in 4 16 30 40 100; do python bench.py --no-cpu-baseline --extra-streams 0 --no-parity --tracks $n 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['tracks'], round(d['ms_per_step']*1e3,1), 'us', round(d['value']))"; done
