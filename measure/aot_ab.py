"""AOT shape family: frame pair under measurement-library switches, interleaved (decode thread groups per band)."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
import siammot_amd.ops as ops
from siammot_amd.config import get_default_cfg
from siammot_amd.emm import EMM
from siammot_amd.structures import BoxList
from siammot_amd.track_utils import build_track_utils
dev = torch.device("cuda", 0)
cfg = get_default_cfg(channels=bench.CHANNELS)
th = cfg.MODEL.TRACK_HEAD
th.POOLER_RESOLUTION, th.SEARCH_REGION, th.PAD_PIXELS = 7, 5.0, 256
th.EMM.USE_CENTERNESS, th.EMM.COSINE_WINDOW_WEIGHT = False, 0.1
image_wh = (bench.NET_HW[1], bench.NET_HW[0])
boxes = bench.synthetic_boxes(30, image_wh)
emm = EMM(cfg, build_track_utils(cfg)).eval()
bench.init_predictor(emm.predictor, boxes)
emm = emm.to(dev)
feats = [bench.synthetic_features(100 + k, dev) for k in range(4)]
det = BoxList(boxes.to(dev), image_wh, mode="xyxy")
det.add_field("ids", torch.arange(30, device=dev))
det.add_field("labels", torch.ones(30, dtype=torch.int64, device=dev))
variants = [dict(kv.split("=") for kv in v.split(",")) if v != "default" else {} for v in sys.argv[1:]] or [{}]
ref = None
for rep in range(2):
    for var in variants:
        with ops.debug_library(**var), torch.no_grad():
            state = emm.extract_cache(feats[3], det)
            for k in range(60):
                z, sr, d = state
                _, result, _ = emm(feats[k % 4], d, sr, template_features=z)
                state = emm.extract_cache(feats[k % 4], det)
            out = result[0].bbox.clone()
            ref = out if ref is None else ref
            torch.cuda.synchronize()
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                for k in range(200):
                    z, sr, d = state
                    _, result, _ = emm(feats[k % 4], d, sr, template_features=z)
                    state = emm.extract_cache(feats[k % 4], det)
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) / 200 * 1e6)
        print(json.dumps({"variant": var or "default", "frame_pair_us": round(min(ts), 1), "same_boxes": bool(torch.equal(out, ref))}), flush=True)
