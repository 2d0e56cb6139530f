"""Host time of the frame-pair API (EMM.forward + EMM.extract_cache, general BoxList interface): bursts of 64 pairs enqueued
without synchronisation, median per pair; with --cprofile the functions that time goes to."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import bench
from siammot_amd.config import get_default_cfg
from siammot_amd.emm import EMM
from siammot_amd.track_utils import build_track_utils
from siammot_amd.structures import BoxList
dev = torch.device("cuda:0")
n = 30
cfg = get_default_cfg(channels=128)
emm = EMM(cfg, build_track_utils(cfg)).to(dev).eval()
feats = [bench.synthetic_features(k, dev) for k in range(4)]
boxes = bench.synthetic_boxes(n, (1280, 704)).to(dev)
det = BoxList(boxes, (1280, 704), mode="xyxy")
det.add_field("ids", torch.arange(n, device=dev)); det.add_field("labels", torch.ones(n, dtype=torch.int64, device=dev))
torch.set_grad_enabled(False)


def step(k, state):
    z, sr, d = state
    _, res, _ = emm(feats[k % 4], d, sr, template_features=z)
    return emm.extract_cache(feats[(k + 1) % 4], d[0])


state = emm.extract_cache(feats[0], det)
for k in range(50): state = step(k, state)
torch.cuda.synchronize()
per = []
for rep in range(20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(64): state = step(k, state)
    per.append((time.perf_counter() - t0) / 64 * 1e6)
    torch.cuda.synchronize()
print("host enqueue us per frame pair: median %.1f, min %.1f" % (float(np.median(per)), min(per)))
if "--cprofile" in sys.argv:
    import cProfile, pstats
    pr = cProfile.Profile()
    for rep in range(30):
        torch.cuda.synchronize()
        pr.enable()
        for k in range(64): state = step(k, state)
        pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(22)
