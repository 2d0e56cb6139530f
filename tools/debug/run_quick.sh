# scratch: one GPU-box session for the kernel under work (edit freely; tools/gpu_round.sh is the full round)
set -x
for s in 2 3 4; do python bench.py --no-cpu-baseline --no-parity --extra-streams $s --steps 500 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['multi_stream'])"; done
