set -x
python -m pytest tests/test_solver.py tests/test_video_results.py -m gpu -q --no-header --tb=short 2>&1 | tail -12
python tools/debug/loop_kernels.py 30 2>&1 | grep tracked
python tools/debug/loop_kernels.py 30 2>&1 | grep tracked
python tools/debug/loop_kernels.py 100 2>&1 | grep tracked
