"""Soak of the order hint's verification (tests/test_hip_parity.py::test_order_hint_is_verified_by_the_head_not_trusted) over
many roi counts: the right hint passes with status 0 and bit-identical results, every kind of wrong hint is reported."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import test_hip_parity as T
import siammot_amd.ops as ops
ops.load_library()
bad = 0
counts = list(range(2, 257, 9)) + [3, 63, 64, 65, 127, 128, 129, 191, 192, 193, 255, 256]
for n in counts:
    try:
        T.test_order_hint_is_verified_by_the_head_not_trusted(ops, n)
    except Exception as e:
        bad += 1
        print("n", n, "FAILED:", type(e).__name__, str(e)[:300], flush=True)
print("hint verification soak done: %d roi counts, failures: %d" % (len(counts), bad))
