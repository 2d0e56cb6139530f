"""Two tracking loops on two torch streams, interleaved frame by frame, against a loop on the default stream: the frame
plan, workspaces, record rings and device-resident pools must not cross (one process serving two videos)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import golden_inputs as gi
from fake_tracker import detections
from siammot_amd.config import get_default_cfg
from siammot_amd.track_head import build_tracking_loop
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 120
dev = torch.device("cuda:0")
cfg = get_default_cfg(channels=32)
cfg.MODEL.TRACK_HEAD.MAX_DORMANT_FRAMES = 3
cfg.MODEL.TRACK_HEAD.TRACK_THRESH = 0.5
cfg.MODEL.TRACK_HEAD.RESUME_TRACK_THRESH = 0.5
torch.manual_seed(5)
loops = [build_tracking_loop(cfg, device=dev, refine_tracks=False) for _ in range(4)]
with torch.no_grad():
    for name in ("cls", "center", "reg"):
        getattr(loops[0].track.tracker.predictor, name).weight.mul_(20.0)
for lp in loops[1:]:
    lp.track.tracker.load_state_dict(loops[0].track.tracker.state_dict())
streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
shapes = gi.feature_shapes((1280, 704), 32)
rs_f = [np.random.RandomState(9), np.random.RandomState(10)]
rs = [np.random.RandomState(5), np.random.RandomState(6), np.random.RandomState(5), np.random.RandomState(6)]
for f in range(frames):
    feats = [tuple(torch.from_numpy(r.standard_normal(s).astype(np.float32)).to(dev) for s in shapes) for r in rs_f]
    torch.cuda.synchronize()
    outs = [None] * 4
    for v in (0, 1):                                    # video v on stream v
        with torch.cuda.stream(streams[v]):
            outs[v] = loops[v](feats[v], detections(rs[v], f % 40).to(dev))
    torch.cuda.synchronize()
    for v in (0, 1):                                    # the same videos one after the other on the default stream
        outs[2 + v] = loops[2 + v](feats[v], detections(rs[2 + v], f % 40).to(dev))
    torch.cuda.synchronize()
    for v in (0, 1):
        a, b = outs[v], outs[2 + v]
        assert torch.equal(a.bbox, b.bbox) and torch.equal(a.get_field("ids"), b.get_field("ids")) and torch.equal(a.get_field("scores"), b.get_field("scores")), (f, v)
print("frames %d x 2 videos on two streams: identical to the default-stream loops; ids started %d / %d" % (
    frames, loops[0].solver.track_pool._max_id + 1, loops[1].solver.track_pool._max_id + 1))
