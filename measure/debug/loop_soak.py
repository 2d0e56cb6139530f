"""Tracking loop soak: memory must stay bounded over hundreds of frames (lazy track cache, workspaces)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import golden_inputs as gi
from fake_tracker import detections
from siammot_amd.config import get_default_cfg
from siammot_amd.track_head import build_tracking_loop
cfg = get_default_cfg(channels=128); cfg.MODEL.TRACK_HEAD.MAX_DORMANT_FRAMES = 5
loop = build_tracking_loop(cfg, device="cuda:0", refine_tracks=False)
rs = np.random.RandomState(0)
shapes = gi.feature_shapes((1280, 704), 128)
feats = [tuple(torch.from_numpy(rs.standard_normal(s).astype(np.float32)).cuda() for s in shapes) for _ in range(2)]
for f in range(400):
    if f == 5:
        loop.solver.start_thresh = 2.0          # no new tracks from here on: the track count can only shrink
        loop.solver.track_thresh = 0.0          # ... and with random weights nothing is ever suspended
    out = loop(feats[f & 1], detections(rs, f % 60).to("cuda:0"))
    if f % 50 == 49:
        torch.cuda.synchronize()
        p = loop.track.track_pool
        print("frame %d: allocated %.1f MB, reserved %.1f MB, active %d dormant %d cache %d started %d" % (
            f + 1, torch.cuda.memory_allocated() / 2**20, torch.cuda.memory_reserved() / 2**20, len(p.get_active_ids()),
            len(p.get_dormant_ids()), len(p.get_cache()), p._max_id + 1), flush=True)
