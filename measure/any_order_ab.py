"""UPPER BOUND on what overlapping the dispatch ramps / tails of the frame pair's four kernels could buy: the same loop with
every launch carrying hipExtAnyOrderLaunch (measurement library, SMOT_ANY_ORDER=1: no barrier between the kernels of the
stream — results are WRONG, kernels of one frame pair and of neighbouring pairs run beside each other).  An empty kernel
measures 4.1 us by its own dispatch timestamps on this chip (profiles/r04_gen4_decomposition.jsonl), four kernels per frame
pair.    python measure/any_order_ab.py 30 100"""
import json, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
import siammot_amd.ops as ops
from siammot_amd.structures import BoxList
from siammot_amd.emm import EMM
from siammot_amd.config import get_default_cfg
from siammot_amd.track_utils import build_track_utils
dev = torch.device("cuda:0")
ops.load_library()
feats = [bench.synthetic_features(k, dev) for k in range(4)]
for n in [int(a) for a in sys.argv[1:]] or [30]:
    boxes = bench.synthetic_boxes(n, (1280, 704)).to(dev)
    det = BoxList(boxes, (1280, 704), mode="xyxy")
    det.add_field("ids", torch.arange(n, device=dev))
    det.add_field("labels", torch.ones(n, dtype=torch.int64, device=dev))
    cfg = get_default_cfg(channels=bench.CHANNELS)
    emm = EMM(cfg, build_track_utils(cfg)).eval()
    bench.init_predictor(emm.predictor, boxes.cpu())
    emm = emm.to(dev)
    for rep in range(2):
        for var in ({}, {"SMOT_ANY_ORDER": "1"}):
            with ops.debug_library(**var), torch.no_grad():
                state = emm.extract_cache(feats[3], det)

                def step(k, state):
                    z, sr, d = state
                    _, res, _ = emm(feats[k % 4], d, sr, template_features=z)
                    return emm.extract_cache(feats[k % 4], det), res
                for k in range(50):
                    state, res = step(k, state)
                torch.cuda.synchronize()
                ts = []
                for _ in range(5):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for k in range(300):
                        state, res = step(k, state)
                    e1.record(); torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1) / 300 * 1e3)
            print(json.dumps({"tracks": n, "variant": var or "ordered", "frame_pair_us": round(min(ts), 2)}), flush=True)
