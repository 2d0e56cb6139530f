"""Round-5 soak of the tracking loop on the device: the default path (frame entry point, early head launch, verified order
hint also over carried dormant rows, steady-frame fast path) against the general path on the SAME random traffic —
detections dropping out, false positives, tracks starting, going dormant (kept 30 frames), resuming, expiring — with the
real HIP head (random weights) in both: every frame's ids / boxes / scores must be bit-identical, no frame may raise (a
false alarm of the hint's verification would), memory must stay bounded.   python measure/debug/loop_soak_r05.py [frames]"""
import json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import golden_inputs as gi
from fake_tracker import detections
import siammot_amd.ops as ops
from siammot_amd.config import get_default_cfg
from siammot_amd.track_head import build_tracking_loop

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
dev = "cuda:0"
torch.set_grad_enabled(False)
cfg = get_default_cfg(channels=128)
cfg.MODEL.TRACK_HEAD.MAX_DORMANT_FRAMES = 30
torch.manual_seed(3)
fast = build_tracking_loop(cfg, device=dev, refine_tracks=False)
slow = build_tracking_loop(cfg, device=dev, refine_tracks=False)
slow.track.tracker.load_state_dict(fast.track.tracker.state_dict())
slow._lean_ok = lambda d: False                     # general path
rs = [np.random.RandomState(0), np.random.RandomState(0)]
g = np.random.RandomState(1)
shapes = gi.feature_shapes((1280, 704), 128)
feats = [tuple(torch.from_numpy(g.standard_normal(s).astype(np.float32)).to(dev) for s in shapes) for _ in range(3)]
ops.FALLBACKS.clear(); ops.SPECULATION.clear(); ops.MEMORY_CARRY.clear()
rows = dormant_max = 0
for f in range(frames):
    if f == frames // 2:                            # second half: a calm scene (nothing starts, nothing is dropped for its score)
        for lp in (fast, slow):
            lp.solver.start_thresh, lp.solver.track_thresh = 2.0, 0.0
    a = fast(feats[f % 3], detections(rs[0], f % 120).to(dev))
    b = slow(feats[f % 3], detections(rs[1], f % 120).to(dev))
    assert torch.equal(a.get_field("ids"), b.get_field("ids")), "ids, frame %d" % f
    assert torch.equal(a.bbox, b.bbox) and torch.equal(a.get_field("scores"), b.get_field("scores")), "frame %d" % f
    assert bool(torch.isfinite(a.bbox).all())
    rows += len(a)
    dormant_max = max(dormant_max, len(fast.solver.track_pool._dormant_ids))
    if f % 250 == 249:
        torch.cuda.synchronize()
        p = fast.solver.track_pool
        print(json.dumps({"frame": f + 1, "allocated_MB": round(torch.cuda.memory_allocated() / 2**20, 1), "active": len(p.get_active_ids()),
                          "dormant": len(p._dormant_ids), "started": p._max_id + 1}), flush=True)
general = ops.FALLBACKS.pop("general_frame", 0)      # (the comparison loop's frames)
print(json.dumps({"frames": frames, "rows": rows, "max_dormant": dormant_max, "identical_to_the_general_path": True,
                  "fallbacks_default_path": dict(ops.FALLBACKS), "general_path_frames_of_the_comparison_loop": general,
                  "heads": dict(ops.SPECULATION), "memory_carry": dict(ops.MEMORY_CARRY)}))
