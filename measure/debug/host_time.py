"""Host time per frame pair (launch loop without waiting for the GPU) vs GPU time per frame pair."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import bench
from siammot_amd import ops
from siammot_amd.config import get_default_cfg
from siammot_amd.emm import EMM
from siammot_amd.structures import BoxList
from siammot_amd.track_utils import build_track_utils
dev = torch.device("cuda:0")
n = 30
image_wh = (1280, 704)
boxes = bench.synthetic_boxes(n, image_wh)
cfg = get_default_cfg(channels=128)
emm = EMM(cfg, build_track_utils(cfg)).eval()
bench.init_predictor(emm.predictor, boxes)
emm = emm.to(dev)
feats = [bench.synthetic_features(100 + k, dev) for k in range(2)]
det = BoxList(boxes.to(dev), image_wh, mode="xyxy")
det.add_field("ids", torch.arange(n, device=dev)); det.add_field("labels", torch.ones(n, dtype=torch.int64, device=dev))
def step(k, state):
    z, sr, d = state
    _, result, _ = emm(feats[k & 1], d, sr, template_features=z)
    return emm.extract_cache(feats[k & 1], det), result
with torch.no_grad():
    state = emm.extract_cache(feats[1], det)
    for k in range(3000): state, _ = step(k, state)
    torch.cuda.synchronize()
    for rep in range(3):
        t0 = time.perf_counter()
        for k in range(2000): state, _ = step(k, state)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("host launch loop %.1f us/step, total %.1f us/step" % ((t1 - t0) / 2000 * 1e6, (t2 - t0) / 2000 * 1e6), flush=True)
    # host-only cost split
    z, sr, d = state
    t0 = time.perf_counter()
    for k in range(2000): emm(feats[k & 1], d, sr, template_features=z)
    t1 = time.perf_counter(); torch.cuda.synchronize()
    t2 = time.perf_counter()
    for k in range(2000): emm.extract_cache(feats[k & 1], det)
    t3 = time.perf_counter(); torch.cuda.synchronize()
    print("host: forward %.1f us, extract_cache %.1f us" % ((t1 - t0) / 2000 * 1e6, (t3 - t2) / 2000 * 1e6))

    # short bursts: the launch queue never fills, so this is the pure host cost
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(40): state, _ = step(k, state)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("burst of 40: host %.1f us/step, total %.1f us/step" % ((t1 - t0) / 40 * 1e6, (t2 - t0) / 40 * 1e6), flush=True)
    import cProfile, pstats
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    for k in range(40): state, _ = step(k, state)
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
