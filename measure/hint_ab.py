"""A/B of the order hint (include/smot_emm.h ``order_hint``), same session, interleaved: the frame-pair loop of bench.py
with the hint (default), with the head ignoring it (SMOT_NO_HINT=1: the extraction still writes it) and with neither
(SMOT_NO_HINT=2 = the round-3 state before the hint).  Per variant: frame pair (torch events around 300 steps, min of
5), the pooling + correlation kernel and the template-pooling launch by the library's event timers.
    python measure/hint_ab.py 30 100"""
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
counts = [int(a) for a in sys.argv[1:]] or [30]
feats = [bench.synthetic_features(k, dev) for k in range(4)]
for n in counts:
    boxes = bench.synthetic_boxes(n, (1280, 704)).to(dev)
    det = BoxList(boxes, (1280, 704), mode="xyxy")
    det.add_field("ids", torch.arange(n, device=dev))
    det.add_field("labels", torch.ones(n, dtype=torch.int64, device=dev))
    cfg = get_default_cfg(channels=bench.CHANNELS)
    emm = EMM(cfg, build_track_utils(cfg)).eval()
    bench.init_predictor(emm.predictor, boxes.cpu())
    emm = emm.to(dev)
    ref = None
    for rep in range(2):
        # ({"SMOT_FUSED_ABL": "6"}: round 5 — the hinted kernel WITHOUT its verification of the hint, i.e. the round-4 kernel)
        for var in ({}, {"SMOT_FUSED_ABL": "6"}, {"SMOT_NO_HINT": "1"}, {"SMOT_NO_HINT": "2"}):
            with ops.debug_library(**var), torch.no_grad():
                state = emm.extract_cache(feats[3], det)

                def step(k, state):
                    z, sr, d = state
                    _, res, _ = emm(feats[k % 4], d, sr, template_features=z)
                    return emm.extract_cache(feats[k % 4], det), res
                for k in range(50):
                    state, res = step(k, state)
                out = torch.cat((res[0].bbox, res[0].get_field("scores")[:, None]), 1).clone()
                ref = out if ref is None else ref
                same = bool(torch.equal(out, ref))
                torch.cuda.synchronize()
                ts = []
                for _ in range(5):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for k in range(300):
                        state, res = step(k, state)
                    e1.record(); torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1) / 300 * 1e3)
                ops.kernel_timer_begin(ops.TIMER_XCORR, 400, 1)
                for k in range(300):
                    state, res = step(k, state)
                torch.cuda.synchronize()
                ms, cnt = ops.kernel_timer_end(ops.TIMER_XCORR)
            print(json.dumps({"tracks": n, "variant": var or "hint", "frame_pair_us": round(min(ts), 2),
                              "fused_us": round(ms * 1e3 / max(cnt, 1), 2), "bitwise_equal": same}), flush=True)
