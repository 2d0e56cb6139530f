"""Round-5 soak of the tracking loop on the device: the default path (frame entry point, early head launch, verified order
hint also over carried dormant rows, steady-frame fast path) against the general path on the SAME traffic, the real HIP
head in both: the reference-generated `longdormant` case's inputs (30-frame dormancy, ~30 active + up to ~110 dormant rows,
tracks starting / going dormant / resuming / expiring all the time) played CYCLES times in a row without a reset, then a
calm stretch (nothing starts, nothing is dropped for its score: the steady-frame path).  Every frame's ids / boxes / scores
must be bit-identical between the two loops, no frame may raise (a false alarm of the hint's verification would), memory
must stay bounded.   python measure/debug/loop_soak_r05.py [cycles]"""
import json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import golden_inputs as gi
import sequence_replay as SR
import siammot_amd.ops as ops
from siammot_amd.track_head import build_tracking_loop

cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = "cuda:0"
torch.set_grad_enabled(False)
inp = gi.SequenceInputs("longdormant")
cfg = SR.sequence_cfg(inp.case)
T = inp.case["frames"]
fast = build_tracking_loop(cfg, device=dev, refine_tracks=False)
slow = build_tracking_loop(cfg, device=dev, refine_tracks=False)
for lp in (fast, slow):
    lp.track.tracker.predictor.load_state_dict({k: torch.from_numpy(v) for k, v in inp.params.items()})
slow._lean_ok = lambda d: False                     # general path
feats = [tuple(torch.from_numpy(f).to(dev) for f in inp.features(t)) for t in range(T)]          # 72 frames, 2.8 GB
ops.FALLBACKS.clear(); ops.SPECULATION.clear(); ops.MEMORY_CARRY.clear()
rows = dormant_max = rows_max = 0
frames = cycles * T + 200
for f in range(frames):
    t = f % T
    if f == cycles * T:                             # the calm stretch
        for lp in (fast, slow):
            lp.solver.start_thresh, lp.solver.track_thresh, lp.solver.resume_track_thresh = 2.0, 0.0, 2.0
    a = fast(feats[t], SR.detections_boxlist(inp, t, dev))
    b = slow(feats[t], SR.detections_boxlist(inp, t, dev))
    assert torch.equal(a.get_field("ids"), b.get_field("ids")), "ids, frame %d" % f
    assert torch.equal(a.bbox, b.bbox) and torch.equal(a.get_field("scores"), b.get_field("scores")), "frame %d" % f
    assert bool(torch.isfinite(a.bbox).all())
    rows += len(a)
    p = fast.solver.track_pool
    dormant_max = max(dormant_max, len(p._dormant_ids))
    rows_max = max(rows_max, len(p._dormant_ids) + len(p.get_active_ids()))
    if f % 72 == 71 or f == frames - 1:
        torch.cuda.synchronize()
        print(json.dumps({"frame": f + 1, "allocated_MB": round(torch.cuda.memory_allocated() / 2**20, 1), "active": len(p.get_active_ids()),
                          "dormant": len(p._dormant_ids), "started": p._max_id + 1}), flush=True)
general = ops.FALLBACKS.pop("general_frame", 0)      # (the comparison loop's frames)
print(json.dumps({"frames": frames, "output_rows": rows, "max_dormant": dormant_max, "max_memory_rows": rows_max,
                  "identical_to_the_general_path": True, "fallbacks_default_path (unhinted_head also counts the comparison loop's heads)":
                  dict(ops.FALLBACKS), "general_path_frames_of_the_comparison_loop": general,
                  "heads": dict(ops.SPECULATION), "memory_carry": dict(ops.MEMORY_CARRY)}))
