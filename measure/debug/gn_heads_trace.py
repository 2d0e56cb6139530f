"""Phase trace (s_memtime ticks, 100 MHz) of tower_gn_heads_kernel<29> — GroupNorm + ReLU + partial heads of the second
yaml family's 29 x 29 towers — on the AOT bench workload (720p, C = 128, 30 tracks).  Per workgroup: start -> channel sums
(the loads of the tile's 16 planes complete here) -> squared deviations -> normalised planes in LDS -> head matrix
instructions -> stores.  Prints mean / max phase ticks, the spread of start and end stamps, and the kernel's span."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import bench
import siammot_amd.ops as ops
from siammot_amd.config import get_default_cfg
from siammot_amd.emm import EMM
from siammot_amd.structures import BoxList
from siammot_amd.track_utils import build_track_utils
dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
cfg = get_default_cfg(channels=bench.CHANNELS)
th = cfg.MODEL.TRACK_HEAD
th.POOLER_RESOLUTION, th.SEARCH_REGION, th.PAD_PIXELS = 7, 5.0, 256
th.EMM.USE_CENTERNESS, th.EMM.COSINE_WINDOW_WEIGHT = False, 0.1
image_wh = (bench.NET_HW[1], bench.NET_HW[0])
boxes = bench.synthetic_boxes(n, image_wh)
emm = EMM(cfg, build_track_utils(cfg)).eval()
bench.init_predictor(emm.predictor, boxes)
emm = emm.to(dev)
feats = [bench.synthetic_features(100 + k, dev) for k in range(2)]
det = BoxList(boxes.to(dev), image_wh, mode="xyxy")
det.add_field("ids", torch.arange(n, device=dev))
det.add_field("labels", torch.ones(n, dtype=torch.int64, device=dev))
lib = ops.load_library()
with torch.no_grad():
    state = emm.extract_cache(feats[1], det)
    z, sr, d = state
    for k in range(20):
        emm(feats[k & 1], d, sr, template_features=z)
    torch.cuda.synchronize()
    grid = (n + 7) // 8 * 8 * 2 * (bench.CHANNELS // 16)
    tr = torch.zeros((65536 + grid) * 8, dtype=torch.int64, device=dev)     # (the head's other kernels stamp the front)
    rows = []
    for rep in range(5):
        tr.zero_()
        lib.smot_debug_trace(ops._ptr(tr))
        emm(feats[rep & 1], d, sr, template_features=z)
        torch.cuda.synchronize()
        lib.smot_debug_trace(ops._ptr(None))
        t = tr.view(-1, 8)[65536:65536 + grid].cpu().numpy()
        t = t[t[:, 5] != 0]
        rows.append(t)
    t = rows[-1]
    ph = np.diff(t[:, :6], axis=1)
    print(json.dumps({"tracks": n, "workgroups_traced": int(t.shape[0]),
                      "phases": ["loads+sums", "squares", "planes", "heads", "stores"],
                      "mean_ticks": [round(float(x), 1) for x in ph.mean(0)], "max_ticks": [int(x) for x in ph.max(0)],
                      "workgroup_ticks_mean": round(float((t[:, 5] - t[:, 0]).mean()), 1),
                      "start_spread_ticks": int(t[:, 0].max() - t[:, 0].min()), "end_spread_ticks": int(t[:, 5].max() - t[:, 5].min()),
                      "kernel_span_ticks": int(t[:, 5].max() - t[:, 0].min()), "tick_ns": 10}))
