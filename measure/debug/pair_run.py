"""The frame-pair loop (EMM.forward + EMM.extract_cache) for rocprofv3 with measurement-library switches:
   python pair_run.py TRACKS [KNOB=VALUE ...]"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import bench
import siammot_amd.ops as ops
from siammot_amd.config import get_default_cfg
from siammot_amd.emm import EMM
from siammot_amd.structures import BoxList
from siammot_amd.track_utils import build_track_utils
dev = torch.device("cuda:0")
n = int(sys.argv[1])
knobs = dict(a.split("=") for a in sys.argv[2:])
cfg = get_default_cfg(channels=128)
boxes = bench.synthetic_boxes(n, (1280, 704))
emm = EMM(cfg, build_track_utils(cfg)).eval(); bench.init_predictor(emm.predictor, boxes); emm = emm.to(dev)
feats = [bench.synthetic_features(k, dev) for k in range(8)]
det = BoxList(boxes.to(dev), (1280, 704), mode="xyxy")
det.add_field("ids", torch.arange(n, device=dev)); det.add_field("labels", torch.ones(n, dtype=torch.int64, device=dev))
torch.set_grad_enabled(False)
with ops.debug_library(**knobs):
    state = emm.extract_cache(feats[0], det)
    for k in range(400):
        z, sr, d = state
        emm(feats[k % 8], d, sr, template_features=z)
        state = emm.extract_cache(feats[(k + 1) % 8], d[0])
    torch.cuda.synchronize()
