# scratch: one GPU-box session for the kernel under work (edit freely; tools/gpu_round.sh is the full round)
set -x
python bench.py --no-cpu-baseline --extra-streams 0 --no-parity --tracks 100 2>&1 | tail -1 | cut -c1-200
python bench.py --no-cpu-baseline --extra-streams 0 --no-parity --tracks 50 --channels 256 --net-hw 1056 1920 --feature-sets 3 2>&1 | tail -1 | cut -c1-200
python bench.py --no-cpu-baseline --extra-streams 0 --no-parity --tracks 4 --net-hw 800 800 2>&1 | tail -1 | cut -c1-200
