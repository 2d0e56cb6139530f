python -m pytest tests -m gpu -q --no-header --tb=short -p no:cacheprovider 2>&1 | tail -15
python -m pytest tests/test_video_results.py tests/test_solver.py -m gpu -q --no-header --tb=short -p no:cacheprovider 2>&1 | tail -3
python -m pytest tests/test_solver.py -m gpu -q --no-header --tb=short -p no:cacheprovider -k lean 2>&1 | tail -3
