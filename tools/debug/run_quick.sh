# scratch: one GPU-box session for the kernel under work (edit freely; tools/gpu_round.sh is the full round)
set -x
python -m pytest tests/test_distributed.py -m gpu -q --no-header --tb=short -x -p no:cacheprovider 2>&1 | tail -12
