"""Why did test_lean_step_equals_general_path see another trajectory on one fresh box?  Prints everything the trajectory
depends on: RNG state, weight checksums, per-frame ids / scores checksums."""
import os, sys, json, hashlib
import numpy as np, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_inputs as gi
from fake_tracker import detections
from siammot_amd.config import get_default_cfg
from siammot_amd.track_head import build_tracking_loop
def h(t): return hashlib.md5(t.detach().cpu().numpy().tobytes()).hexdigest()[:10]
dev = torch.device("cuda:0")
out = {"initial_seed": torch.initial_seed()}
cfg = get_default_cfg(channels=32); cfg.MODEL.TRACK_HEAD.MAX_DORMANT_FRAMES = 3
cfg.MODEL.TRACK_HEAD.TRACK_THRESH = 0.5; cfg.MODEL.TRACK_HEAD.RESUME_TRACK_THRESH = 0.5
loop = build_tracking_loop(cfg, device=dev, refine_tracks=False)
with torch.no_grad():
    for name in ("cls", "center", "reg"): getattr(loop.track.tracker.predictor, name).weight.mul_(20.0)
out["weights"] = h(torch.cat([p.flatten() for p in loop.track.tracker.parameters()]))
shapes = gi.feature_shapes((1280, 704), 32)
rs_f = np.random.RandomState(9); rs = np.random.RandomState(5)
frames = []
for f in range(16):
    feats = tuple(torch.from_numpy(rs_f.standard_normal(s).astype(np.float32)).to(dev) for s in shapes)
    o = loop(feats, detections(rs, f).to(dev))
    frames.append([h(o.bbox), h(o.get_field("scores")), sorted(o.get_field("ids").tolist())[-3:], len(o)])
out["frames"] = frames
out["kill_ids"] = sorted(loop.solver.track_pool._kill_ids)
print(json.dumps(out))
