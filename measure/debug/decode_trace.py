"""Phase trace (s_memtime ticks) of the one-launch decode kernel inside the one-call head, bench workload."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench
import siammot_amd.ops as ops
from siammot_amd.config import get_default_cfg
from siammot_amd.emm import EMM
from siammot_amd.structures import BoxList
from siammot_amd.track_utils import build_track_utils
dev = torch.device("cuda:0")
lib = ops.load_library()
for n in [int(a) for a in sys.argv[1:]] or [30]:
    image_wh = (1280, 704)
    boxes = bench.synthetic_boxes(n, image_wh)
    cfg = get_default_cfg(channels=128)
    emm = EMM(cfg, build_track_utils(cfg)).eval(); bench.init_predictor(emm.predictor, boxes); emm = emm.to(dev)
    feats = bench.synthetic_features(1, dev)
    det = BoxList(boxes.to(dev), image_wh); det.add_field("ids", torch.arange(n, device=dev)); det.add_field("labels", torch.ones(n, dtype=torch.int64, device=dev))
    with torch.no_grad():
        z, sr, d = emm.extract_cache(feats, det)
        logits = None
        resp = ops.sr_xcorr_fused(feats, d[0].bbox, sr[0].bbox, z, 30, 15, (0.25, 0.125, 0.0625, 0.03125), 2, 512)
        lg = ops.emm_predictor(resp, emm.predictor.param_dict())
        f = lambda: ops.emm_decode(lg, sr[0].bbox, d[0].bbox, 30, 15, 512, clip_wh=image_wh)
        for _ in range(20): f()
        torch.cuda.synchronize()
        tr = torch.zeros(n * 17 * 8, dtype=torch.int64, device=dev)
        lib.smot_debug_trace(ops._ptr(tr)); f(); torch.cuda.synchronize(); lib.smot_debug_trace(ops._ptr(None))
    t = tr.view(n, 17, 8).cpu().numpy().astype(np.float64)
    d7 = np.diff(t[:, :, :7], axis=2).reshape(-1, 6)
    last = t[:, :, 7].max(axis=1) - t[:, :, 6].max(axis=1)
    print(json.dumps({"tracks": n, "phase_ticks_mean(load+tables,walk,reduce,exact,reduce2,publish+ticket)": [round(float(x)) for x in d7.mean(0)],
                      "phase_max": [int(x) for x in d7.max(0)], "finalize_after_last_ticket_mean": round(float(last.mean())),
                      "span": int(t[:, :, 7].max() - t[:, :, 0].min()), "start_spread": int(t[:, :, 0].max() - t[:, :, 0].min())}))
