"""Soak: the seeded geometry fuzz of tests/test_hip_parity.py over many seeds (not part of the suite)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import test_hip_parity as T
import siammot_amd.ops as ops
ops.load_library()
bad = 0
for seed in range(3, int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    try:
        T.test_emm_random_geometry_against_oracle(ops, seed)
    except AssertionError as e:
        bad += 1
        print("seed", seed, "FAILED:", str(e)[:300], flush=True)
print("soak done, failures:", bad)
