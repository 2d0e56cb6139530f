"""Soak: the order-hint test of tests/test_hip_parity.py over many roi counts (seeds follow the count)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import test_hip_parity as T
import siammot_amd.ops as ops
ops.load_library()
bad = 0
counts = list(range(2, 258, 5)) + [63, 64, 65, 127, 128, 129, 191, 192, 193, 255, 256]
for n in counts:
    try:
        T.test_order_hint_lists_the_rois_in_cost_order_and_changes_nothing(ops, n)
    except AssertionError as e:
        bad += 1
        print("n", n, "FAILED:", str(e)[:300], flush=True)
print("hint soak done: %d roi counts, failures: %d" % (len(counts), bad))
