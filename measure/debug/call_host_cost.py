"""Host cost (enqueue only) of the bindings the tracking loop calls per frame: 300 back-to-back calls, queue drained before."""
import os, sys, time, json
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench
import siammot_amd.ops as ops
from siammot_amd.box_refine import build_refine_tracks
from siammot_amd.config import get_default_cfg
dev = torch.device("cuda:0")
feats = bench.synthetic_features(100, dev)
n = 30
boxes = bench.synthetic_boxes(n, (1280, 704)).to(dev)
cfg = get_default_cfg(channels=128)
refine = build_refine_tracks(cfg, 128); refine.box.to(dev).eval()
ids = torch.arange(n, device=dev); labels = torch.ones(n, dtype=torch.int64, device=dev); conf = torch.rand(n, device=dev)
scales = (0.25, 0.125, 0.0625, 0.03125)
x = torch.randn(n, 6272, device=dev); w = torch.randn(1024, 6272, device=dev)
calls = {
    "refine_raw (8 launches)": lambda: refine.refine_raw(feats, boxes, conf, ids, labels, (1280, 704)),
    "roi_align_levels 7x7 (1 launch)": lambda: ops.roi_align_levels(feats, boxes, boxes, 7, scales, 2),
    "linear_rows (2 launches)": lambda: ops.linear_rows(x, w),
    "search_region (1 launch)": lambda: ops.search_region(boxes, 512, 1.0, 0),
    "torch.addmm (1 launch)": lambda: torch.mm(x, w.t()),
}
for name, f in calls.items():
    for _ in range(20): f()
    torch.cuda.synchronize()
    ts = []
    for rep in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100): f()
        ts.append((time.perf_counter() - t0) / 100 * 1e6)
        torch.cuda.synchronize()
    print(json.dumps({"call": name, "host_us_min": round(min(ts), 1), "host_us_median": round(sorted(ts)[2], 1)}), flush=True)
