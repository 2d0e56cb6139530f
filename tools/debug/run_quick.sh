python -m pytest tests/test_distributed.py -m gpu -q --no-header --tb=short -p no:cacheprovider 2>&1 | tail -5
