"""Soak: the tracking loop's three paths (frame entry point in two calls = default, Python-composed one-launch path,
general path) on the same 300-frame sequence with tracks going dormant, resuming and expiring — outputs, memory and
pool must be identical frame by frame.  usage: [frames]"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import golden_inputs as gi
from fake_tracker import detections
from siammot_amd.config import get_default_cfg
from siammot_amd.track_head import build_tracking_loop
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda:0")
cfg = get_default_cfg(channels=32)
cfg.MODEL.TRACK_HEAD.MAX_DORMANT_FRAMES = int(os.environ.get("MAXD", "3"))      # (MAXD=30: the MOT17 yaml's value)
cfg.MODEL.TRACK_HEAD.TRACK_THRESH = 0.5
cfg.MODEL.TRACK_HEAD.RESUME_TRACK_THRESH = 0.5
torch.manual_seed(int(os.environ.get("SEED", "0")))
refine = os.environ.get("REFINE") is not None      # box-head refinement on: frame entry point vs Python-composed only (the
                                                   # general path's BoxList refinement agrees to rounding, not to the bit)
if refine:
    from siammot_amd.box_refine import build_refine_tracks
    cfg.MODEL.ROI_BOX_HEAD.MLP_HEAD_DIM = 128
    heads = [build_refine_tracks(cfg, 32) for _ in range(2)]
    heads[1].box.load_state_dict(heads[0].box.state_dict())
    loops = [build_tracking_loop(cfg, device=dev, refine_tracks=h) for h in heads]
    for h in heads:
        h.box.to(dev).eval()
else:
    loops = [build_tracking_loop(cfg, device=dev, refine_tracks=False) for _ in range(3)]
with torch.no_grad():
    for name in ("cls", "center", "reg"):
        getattr(loops[0].track.tracker.predictor, name).weight.mul_(20.0)
for lp in loops[1:]:
    lp.track.tracker.load_state_dict(loops[0].track.tracker.state_dict())
loops[1].native_frame = False
for lp in loops:
    lp.lazy_memory = os.environ.get("NO_LAZY") is None
    lp.solver.track_pool.mirror_skip = os.environ.get("NO_SKIP") is None
if not refine:
    loops[2]._lean_ok = lambda d: False
shapes = gi.feature_shapes((1280, 704), 32)
rs_f = np.random.RandomState(9)
rs = [np.random.RandomState(5) for _ in loops]
taken = [0, 0]
n0, n1 = loops[0]._step_native, loops[1]._step_lean
loops[0]._step_native = lambda f, d, next_features=None: (taken.__setitem__(0, taken[0] + 1), n0(f, d, next_features=next_features))[1]
loops[1]._step_lean = lambda f, d: (taken.__setitem__(1, taken[1] + 1), n1(f, d))[1]
dormant_frames = kills = 0
switch = os.environ.get("SWITCH") is not None      # loop 0 changes its path at random from frame to frame
rs_sw = np.random.RandomState(77)
lean_ok0, dev_path0 = loops[0]._lean_ok, loops[0].solver._device_path
modes = [0, 0, 0, 0]
for f in range(frames):
    if switch:
        mode = int(rs_sw.randint(0, 4))                # 0 frame entry point, 1 Python-composed, 2 general + device solver, 3 general + host solver
        modes[mode] += 1
        loops[0].native_frame = mode == 0
        loops[0]._lean_ok = lean_ok0 if mode < 2 else (lambda d: False)
        loops[0].solver._device_path = dev_path0 if mode < 3 else (lambda *a, **k: False)
    if f == 0:
        feats_next = tuple(torch.from_numpy(rs_f.standard_normal(s).astype(np.float32)).to(dev) for s in shapes)
    feats = feats_next
    feats_next = tuple(torch.from_numpy(rs_f.standard_normal(s).astype(np.float32)).to(dev) for s in shapes)
    if os.environ.get("AHEAD") is not None:      # loop 0 is shown the next frame's features (speculative next-frame head)
        outs = [loops[0](feats, detections(rs[0], f % 40).to(dev), next_features=feats_next)] + \
               [lp(feats, detections(r, f % 40).to(dev)) for lp, r in zip(loops[1:], rs[1:])]
    else:
        outs = [lp(feats, detections(r, f % 40).to(dev)) for lp, r in zip(loops, rs)]
    peek = int(os.environ.get("PEEK", "1"))        # memories compared every PEEK-th frame (looking at one builds it, and a
                                                   # built memory takes no speculative head: PEEK=5 with AHEAD leaves most
                                                   # frames to the speculation, dormant rows copied ahead included)
    for k in range(1, len(loops)):
        a, b = outs[0], outs[k]
        assert torch.equal(a.bbox, b.bbox) and torch.equal(a.get_field("ids"), b.get_field("ids")), (f, k)
        assert torch.equal(a.get_field("scores"), b.get_field("scores")), (f, k)
        if f % peek:
            continue
        ma, mb = loops[0].track_memory, loops[k].track_memory
        if not (ma[0].shape == mb[0].shape and torch.equal(ma[0], mb[0]) and torch.equal(ma[1][0].bbox, mb[1][0].bbox) and torch.equal(ma[2][0].bbox, mb[2][0].bbox)):
            ids_a, ids_b = ma[2][0].get_field("ids").cpu().tolist(), mb[2][0].get_field("ids").cpu().tolist()
            print("MISMATCH frame", f, "loop", k, "rows", ma[0].shape[0], mb[0].shape[0], "taken", taken, "ids equal", ids_a == ids_b)
            if ma[0].shape == mb[0].shape:
                dz = (ma[0] - mb[0]).abs().amax(dim=(1, 2, 3)).cpu()
                bad = torch.nonzero(dz > 0).flatten().tolist()
                act = loops[0].solver.track_pool.get_active_ids()
                print("  template rows differing:", bad[:20], "ids", [ids_a[i] for i in bad[:20]], "active?", [ids_a[i] in act for i in bad[:20]], "max", float(dz.max()))
                dsr = (ma[1][0].bbox - mb[1][0].bbox).abs().amax(dim=1).cpu()
                dbx = (ma[2][0].bbox - mb[2][0].bbox).abs().amax(dim=1).cpu()
                print("  sr rows differing:", torch.nonzero(dsr > 0).flatten().tolist()[:20], "box rows differing:", torch.nonzero(dbx > 0).flatten().tolist()[:20])
                print("  n_det", len(detections(np.random.RandomState(5), f % 40)), "nan in a/b:", bool(torch.isnan(ma[0]).any()), bool(torch.isnan(mb[0]).any()))
            sys.exit(1)
        assert torch.equal(ma[2][0].get_field("ids"), mb[2][0].get_field("ids")), (f, k)
        for fld in ("labels", "scores"):
            assert torch.equal(ma[2][0].get_field(fld), mb[2][0].get_field(fld)), (f, k, fld)
            assert torch.equal(ma[1][0].get_field(fld), mb[1][0].get_field(fld)), (f, k, fld)
        pa, pb = loops[0].solver.track_pool, loops[k].solver.track_pool
        assert pa.get_active_ids() == pb.get_active_ids() and pa._dormant_ids == pb._dormant_ids and pa._max_id == pb._max_id, (f, k)
    if os.environ.get("WATCH") and f >= int(os.environ.get("WATCH_FROM", "138")):
        wid = int(os.environ["WATCH"])
        for k, lp in enumerate(loops):
            m = lp.track_memory
            ids_m = m[2][0].get_field("ids").cpu().tolist()
            row = ids_m.index(wid) if wid in ids_m else -1
            p = lp.solver.track_pool
            ce = p._cache.get(wid)
            print("f", f, "loop", k, "rows", len(ids_m), "row", row, "zsum", float(m[0][row].double().sum()) if row >= 0 else None,
                  "box", m[2][0].bbox[row].cpu().tolist() if row >= 0 else None,
                  "active", wid in p.get_active_ids(), "dormant", p._dormant_ids.get(wid), "cache", None if ce is None else float(ce[0].double().sum()),
                  "pending", p._pending is not None, "taken", list(taken))
    dormant_frames += bool(loops[0].solver.track_pool.get_dormant_ids())
if switch:
    print("path switching: frames per mode (entry point, composed, general+device solver, general+host solver):", modes)
import siammot_amd.ops as _ops
print("speculative heads:", dict(_ops.SPECULATION), "dormant rows copied on the device:", dict(_ops.MEMORY_CARRY),
      "frames that concatenated them on the host:", _ops.FALLBACKS["dormant_rows_on_the_host"])
print("frames %d: identical on all three paths; native frames %d, lean frames %d, frames with dormant tracks %d, ids started %d, killed %d"
      % (frames, taken[0], taken[1], dormant_frames, loops[0].solver.track_pool._max_id + 1, len(loops[0].solver.track_pool._kill_ids)))
