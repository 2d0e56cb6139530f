set -x
python -m pytest tests/test_hip_parity.py -m gpu -q --no-header --tb=short -x -k "fused or pool or emm or roi or levels" 2>&1 | tail -3
python tools/debug/pairing_probe.py 30 100 2>&1 | grep "^{"
