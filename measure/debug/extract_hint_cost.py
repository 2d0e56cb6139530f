"""Template extraction (EMM.extract_cache's one launch) with and without the order-hint writer's extra workgroups: run under
rocprofv3 --kernel-trace --stats; the two phases are separated by 50 launches of smot::empty_kernel... (here: by call count)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import bench
import siammot_amd.ops as ops
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
feats = [bench.synthetic_features(k, dev) for k in range(4)]
boxes = bench.synthetic_boxes(n, (1280, 704)).to(dev)
scales = (0.25, 0.125, 0.0625, 0.03125)
for hint in (True, False):
    f = lambda i: ops.emm_extract_cache(feats[i % 4], boxes, 15, scales, 2, 512.0, 1.0, 0.0, hint=hint)
    for i in range(30): f(i)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for rep in range(5):
        ev0.record()
        for i in range(200): f(i)
        ev1.record(); torch.cuda.synchronize()
        best = min(best, ev0.elapsed_time(ev1) / 200 * 1e3)
    print("hint=%s: %.2f us per call (back-to-back launches, event bracket over 200)" % (hint, best))
