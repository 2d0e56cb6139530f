mkdir -p gpurun_out
for i in 1 2; do python bench.py --no-cpu-baseline --no-kernel-timer 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('notimer', d['ms_per_step'])"; done
for i in 1 2; do python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('timer', d['ms_per_step'], d['roofline']['avg_launch_us'])"; done
python bench.py --no-cpu-baseline --no-kernel-timer --steps 2000 --warmup 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('notimer 2000 steps', d['ms_per_step'])"
python bench.py --no-cpu-baseline --no-kernel-timer --tracks 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('notimer N=100', d['ms_per_step'])"
