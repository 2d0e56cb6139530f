"""Second yaml family in the tracking loop (general path, device solver): the masked 7x7 template extraction enqueued before
the frame's synchronisation vs the extraction after it — identical outputs and memories over 40 frames."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import golden_inputs as gi
from fake_tracker import detections
from siammot_amd.config import get_default_cfg
from siammot_amd.track_head import build_tracking_loop
dev = torch.device("cuda:0")
cfg = get_default_cfg(channels=32)
th = cfg.MODEL.TRACK_HEAD
th.POOLER_RESOLUTION, th.SEARCH_REGION, th.PAD_PIXELS = 7, 5.0, 256
th.EMM.USE_CENTERNESS, th.EMM.COSINE_WINDOW_WEIGHT = False, 0.1
th.MAX_DORMANT_FRAMES, th.TRACK_THRESH, th.RESUME_TRACK_THRESH = 3, 0.5, 0.5
torch.manual_seed(2)
loops = [build_tracking_loop(cfg, device=dev, refine_tracks=False) for _ in range(2)]
with torch.no_grad():
    for name in ("cls", "center", "reg"):
        getattr(loops[0].track.tracker.predictor, name).weight.mul_(20.0)
loops[1].track.tracker.load_state_dict(loops[0].track.tracker.state_dict())
loops[1].track.tracker.extract_cache_rows = lambda *a, **k: None          # extraction after the synchronisation
shapes = gi.feature_shapes((1280, 704), 32)
rs_f = np.random.RandomState(9)
rs = [np.random.RandomState(5), np.random.RandomState(5)]
for f in range(40):
    feats = tuple(torch.from_numpy(rs_f.standard_normal(s).astype(np.float32)).to(dev) for s in shapes)
    a, b = [lp(feats, detections(r, f).to(dev)) for lp, r in zip(loops, rs)]
    assert torch.equal(a.bbox, b.bbox) and torch.equal(a.get_field("ids"), b.get_field("ids")), f
    ma, mb = loops[0].track_memory, loops[1].track_memory
    assert torch.equal(ma[0], mb[0]) and torch.equal(ma[1][0].bbox, mb[1][0].bbox) and torch.equal(ma[2][0].bbox, mb[2][0].bbox), f
print("AOT loop: 40 frames identical; rz", loops[0].track.tracker.rz, "rx", loops[0].track.tracker.rx, "memory rows", ma[0].shape[0],
      "ids started", loops[0].solver.track_pool._max_id + 1)
