import os, sys, time, cProfile, pstats, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import bench
from siammot_amd.config import get_default_cfg
from siammot_amd.structures import BoxList
from siammot_amd.track_head import build_tracking_loop
dev = torch.device("cuda:0"); n = 30; image_wh = (1280, 704)
boxes = bench.synthetic_boxes(n, image_wh).to(dev)
feats = [bench.synthetic_features(100 + k, dev) for k in range(2)]
loop = build_tracking_loop(get_default_cfg(channels=128), device=dev, refine_tracks=False)
bench.init_predictor(loop.track.tracker.predictor, boxes.cpu()); loop.track.tracker.to(dev)
def mk(k):
    d = BoxList(boxes + float(k & 1), image_wh, mode="xyxy")
    d.add_field("ids", torch.full((n,), -1, dtype=torch.int64, device=dev)); d.add_field("labels", torch.ones(n, dtype=torch.int64, device=dev)); d.add_field("scores", torch.full((n,), 0.9, device=dev))
    return d
_pre = [mk(0), mk(1)]
def dets(k):          # the detector's output exists already: a fresh BoxList object around resident tensors
    p = _pre[k & 1]
    d = BoxList(p.bbox, image_wh, mode="xyxy")
    d.add_field("ids", p.get_field("ids")); d.add_field("labels", p.get_field("labels")); d.add_field("scores", p.get_field("scores").clone())
    return d
for k in range(50): loop(feats[k & 1], dets(k))
torch.cuda.synchronize(); t0 = time.perf_counter()
for k in range(300): loop(feats[k & 1], dets(k))
torch.cuda.synchronize(); print("loop: %.0f us per frame" % ((time.perf_counter() - t0) / 300 * 1e6))
pr = cProfile.Profile(); pr.enable()
for k in range(100): loop(feats[k & 1], dets(k))
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(40)
